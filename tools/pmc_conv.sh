#!/bin/bash
# SQ counters of the conv3x3 kernels (run on the GPU box through gpurun); prints per-kernel means.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $O/$tag --output-format csv -- python $R/tools/experiments/exp_conv3x3_time.py > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n): return n.replace("void hdn::cv::conv3x3_kernel<hdn::cv::Cfg<", "conv<")[:44]
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    if "conv<" not in k: continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%s  dur %.1f us  waves %d" % (k, sum(dur[k]) / len(dur[k]), c.get("SQ_WAVES", 0)))
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"):
        print("    %-26s %14.0f  %5.1f %% of wave cycles" % (n, c.get(n, 0), 100 * c.get(n, 0) / wc))
    for n in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_LDS"):
        print("    %-26s %14.0f" % (n, c.get(n, 0)))
PY
find $O -name "*.csv" -delete     # the raw traces exceed what gpurun copies back; the table above is the product
