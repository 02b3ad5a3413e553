#!/bin/bash
# Effective shader clock of the north-star kernels: GRBM_GUI_ACTIVE (busy cycles of the graphics clock domain) over the
# kernel's duration, per variant.  Output: gpurun_out/pmc_clock/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_clock
rm -rf $O; mkdir -p $O
for v in fft fft2w direct; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/$v --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --only-north --north $v --no-cpu-baseline --no-full-head > $O/$v.log 2>&1
done
python - <<PY > $O/summary.txt
import csv, glob, collections
for v in ("fft", "fft2w", "direct"):
    acc = collections.defaultdict(list); dur = {}
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "xcorr_north" in row["Kernel_Name"]:
                acc[(row["Dispatch_Id"], row["Counter_Name"])].append(float(row["Counter_Value"]))
                dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3 if "End_Timestamp" in row else None
    for f in glob.glob("$O/%s/**/*kernel_trace.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "xcorr_north" in row["Kernel_Name"]:
                dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    ids = sorted({d for d, _ in acc}, key=int)[-10:]
    for d in ids[-3:]:
        c = {n: sum(vals) for (dd, n), vals in acc.items() if dd == d}
        us = dur.get(d)
        print(v, "dispatch", d, "dur %.1f us" % us, {k: int(x) for k, x in c.items()},
              "GUI_ACTIVE/dur = %.0f MHz" % (c.get("GRBM_GUI_ACTIVE", 0) / us), "wave_quads/wave*4/dur = %.0f MHz" % (c.get("SQ_WAVE_CYCLES", 0) / max(1, c.get("SQ_WAVES", 1)) * 4 / us))
PY
cat $O/summary.txt
