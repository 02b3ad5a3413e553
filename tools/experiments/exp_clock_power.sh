#!/bin/bash
# Which regime does the north-star kernel run in?  Samples the GPU's shader clock and socket power (rocm-smi / sysfs) every
# 50 ms while each variant of the 31x31 (x) 61x61 kernel runs back to back for ~3 s, on zeros and on random data.
# Output: gpurun_out/clock_power.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/clock_power.txt
: > $O
rocm-smi --showmaxpower --showclocks --showpower --showperflevel >> $O 2>&1
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $HW" >> $O
ls $HW >> $O 2>&1
sample() {  # $1 = tag
  while [ -f /tmp/sampling ]; do
    p=$(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null)
    f=$(cat $HW/freq1_input 2>/dev/null)
    echo "$1 power_uW=$p sclk_Hz=$f" >> $O
    sleep 0.05
  done
}
for v in fft fft2w direct; do
  for d in zeros random; do
    touch /tmp/sampling
    sample "$v/$d" &
    SP=$!
    python - "$v" "$d" >> $O 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from hdn_amd import xcorr as X
v, d = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
mk = (lambda s: torch.zeros(s, device=dev)) if d == "zeros" else (lambda s: torch.relu(torch.randn(s, device=dev)))
x, k = mk((64, 256, 61, 61)), mk((64, 256, 31, 31))
with X.north_variant(v):
    for _ in range(200): X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): X.xcorr_depthwise(x, k)
        e1.record(); torch.cuda.synchronize(); n += 1
        last = e0.elapsed_time(e1) * 1000 / 200
    print(f"RESULT {v}/{d}: {last:.1f} us per launch (last block of 200)")
PY
    rm -f /tmp/sampling
    wait $SP
  done
done
python - >> $O <<'PY'
import re, collections
rows = collections.defaultdict(list)
for line in open("gpurun_out/clock_power.txt"):
    m = re.match(r"(\S+) power_uW=(\d*) sclk_Hz=(\d*)", line)
    if m and m.group(2) and m.group(3):
        rows[m.group(1)].append((int(m.group(2)) / 1e6, int(m.group(3)) / 1e6))
print("\n# summary over the last 2/3 of each run's samples: mean power W, mean sclk MHz, min sclk, max sclk")
for k, v in rows.items():
    v = v[len(v) // 3:]
    print(f"{k:16s} {sum(a for a, _ in v) / len(v):8.1f} W  {sum(b for _, b in v) / len(v):8.1f} MHz  [{min(b for _, b in v):.0f}, {max(b for _, b in v):.0f}]  n={len(v)}")
PY
grep -v "power_uW" $O | tail -40
