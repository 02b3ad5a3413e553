#!/bin/bash
# Matrix-pipe occupancy of the round-5 convolution kernels (conv3x3_v2_kernel, conv3x3s2_v2_kernel, trunk_stem_mfma_kernel) from SQ counters; run on
# the GPU box through gpurun.  Prints per kernel: duration, MFMA-busy cycles against busy / wave cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv_v2
rm -rf $O; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/$tag --output-format csv -- python $R/tools/experiments/exp_full_head_graph.py > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n):
    n = n.replace("void hdn::cv::", "").replace("void hdn::cvs::", "").replace("hdn::cvs::", "").replace("hdn::cv::", "").replace("void hdn::stem_mc::", "").replace("hdn::stem_mc::", "")
    return n[:56]
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    if "conv3x3" not in k and "stem" not in k: continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    d = sum(dur[k]) / max(1, len(dur[k]))
    print("%-56s dur %5.1f us  waves %5d  MFMA insts %8d  MFMA-busy %11.0f = %4.1f %% of SQ_BUSY_CU_CYCLES (%11.0f), %4.1f %% of wave cycles / 4 (%11.0f)" % (
        k, d, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_MFMA", 0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0),
        100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, c.get("SQ_BUSY_CU_CYCLES", 0)), c.get("SQ_BUSY_CU_CYCLES", 0),
        100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0) / 4), c.get("SQ_WAVE_CYCLES", 0)))
PY
find $O -name "*kernel_trace.csv" -delete
