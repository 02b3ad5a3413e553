#!/bin/bash
# The 1 / 2 / 4 / 8-GPU line of BASELINE.json in one command, on an 8-GPU MI355X node:
#     tools/scale_sweep.sh [steps] [warmup]      -> gpurun_out/scale/*.json + a table on stdout
# Runs bench.py --gpus N (kernels, configs[1]) and --workload full (configs[2] per GPU) for N = 1, 2, 4, 8 with the RCCL all-gather
# of the C ABI, and for N > 1 again with the direct-write one-shot gather (falls back to RCCL, with a warning in the log, when a
# pair of GPUs has no peer access).  Weak scaling: 64 pairs per GPU; efficiency = value(N) / (N * value(1)).
# UNMEASURED ON HARDWARE until an 8-GPU node runs this: the build boxes have one GPU.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-100}; WARM=${2:-10}
OUT=gpurun_out/scale; mkdir -p $OUT
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
port=29540
run() {  # n tag extra-args...
  local n=$1 tag=$2; shift 2
  if [ "$n" -gt "$NG" ]; then echo "skip $tag: $n GPUs asked, $NG present" >&2; return; fi
  port=$((port + 1))
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-sequence --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.log
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --steps $STEPS --warmup $WARM --no-sequence --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.log
  fi
  echo "$tag rc=$?" >&2
}
for n in 1 2 4 8; do
  run $n kernels_rccl_$n
  run $n full_rccl_$n --workload full
  if [ $n -gt 1 ]; then
    run $n kernels_oneshot_$n --collective oneshot
    run $n full_oneshot_$n --workload full --collective oneshot
  fi
done
python - <<'PY'
import glob, json, os, re
rows = {}
for f in sorted(glob.glob("gpurun_out/scale/*.json")):
    m = re.match(r"(kernels|full)_(rccl|oneshot)_(\d+)\.json", os.path.basename(f))
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception:
        continue
    rows[(m.group(1), m.group(2), int(m.group(3)))] = d
print(f"{'workload':8s} {'collective':10s} {'gpus':>4s} {'frames/s':>12s} {'ms/step':>9s} {'efficiency':>10s}")
for (w, c, n), d in sorted(rows.items()):
    base = rows.get((w, "rccl", 1))
    eff = d["value"] / (n * base["value"]) if base else float("nan")
    print(f"{w:8s} {c:10s} {n:4d} {d['value']:12.0f} {d['ms_per_step']:9.4f} {eff:10.3f}")
PY
