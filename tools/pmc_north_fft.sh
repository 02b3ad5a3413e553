#!/bin/bash
# SQ counters of the FFT north kernel (run on the GPU box through gpurun); writes gpurun_out/pmc_fft/*.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HDN_NORTH_FFT=${HDN_NORTH_FFT:-2}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_MFMA SQ_IFETCH SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_fft/$tag --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --only-north --no-cpu-baseline > $R/gpurun_out/pmc_fft/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_fft/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "xcorr_north" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
# one row per (dispatch, counter, dimension instance?) -> sum per dispatch is what rocprofv3 prints; average over dispatches
for (k, c), v in sorted(acc.items()):
    print("%-42s %-24s n=%-4d mean %16.0f" % (k, c, len(v), sum(v) / len(v)))
PY
