#!/bin/bash
# Instruction-cache counters of the north-star kernel variants (run through gpurun).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ic
rm -rf $O; mkdir -p $O
for v in fft fft2w; do
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/$v --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown --only-north --north $v > $O/$v.log 2>&1
done
python - <<PY
import csv, glob, collections
for v in ("fft", "fft2w"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "xcorr_north" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(v, {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
