#!/bin/bash
# SQ counters of every kernel of one bench step (run on the GPU box through gpurun); prints per-kernel means.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
rm -rf $O; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_F32 SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $O/$tag --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown ${BENCH_ARGS} > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"][:48]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    if "hdn::" not in k: continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%s  dur %.1f us  waves %d" % (k, sum(dur[k]) / len(dur[k]), c.get("SQ_WAVES", 0)))
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
        print("    %-22s %14.0f  %5.1f %% of wave cycles" % (n, c.get(n, 0), 100 * c.get(n, 0) / wc))
    for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        print("    %-22s %14.0f" % (n, c.get(n, 0)))
PY
