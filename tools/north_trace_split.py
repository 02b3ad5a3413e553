#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of `python bench.py` (tools/profile_round.sh): the 31x31 (x) 61x61 kernel's launches split into the ones INSIDE a step (the next
launch on the device is the 13x13 kernel of the same step) and the back-to-back ones (pre-warm, the `sustained` graph, the clock sampler's replays), because
rocprofv3's own --stats average mixes them.  usage: north_trace_split.py <directory holding *kernel_trace.csv>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
in_step, b2b = [], []
for i, (s, e, n) in enumerate(ks):
    if "xcorr_north_fft4" not in n:
        continue
    nxt = [x[2] for x in ks[i + 1:i + 4]]
    (in_step if any("circ13" in x for x in nxt) else b2b).append((e - s) / 1e3)
q = lambda v: "n = %5d   mean %6.2f us   median %6.2f   min %6.2f   max %6.2f" % (len(v), sum(v) / len(v), sorted(v)[len(v) // 2], min(v), max(v)) if v else "none"
B = 369819648
print("xcorr_north_fft4_kernel<4>, kernel durations from the rocprofv3 kernel trace of `python bench.py --no-cpu-baseline --no-sequence` (algorithmic bytes per launch %d)" % B)
print("  inside a step (followed by the step's 13x13 launch): ", q(in_step), "  -> %.3f of 8 TB/s" % (B / (sum(in_step) / len(in_step) * 1e-6) / 8e12) if in_step else "")
print("  back to back (pre-warm, sustained graph, clock sampler): ", q(b2b), "  -> %.3f of 8 TB/s" % (B / (sum(b2b) / len(b2b) * 1e-6) / 8e12) if b2b else "")
