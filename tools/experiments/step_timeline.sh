# Timeline of the configs[1] step: rocprofv3 kernel trace of a short bench run, the last steps' kernels in time order with their queue.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/step_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --no-sequence --no-full-head --roofline-steps 2 ${BENCH_EXTRA} > $O/log.txt 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ks.sort()
# the timed region: find the north launches, take the 12th..15th of the last 22
north = [i for i, k in enumerate(ks) if "xcorr_north_fft4" in k[2] and any("circ13" in x[2] for x in ks[i + 1:i + 6])]    # launches of full steps only
i0, i1 = north[-6], north[-3]
t0 = ks[i0][0]
for s, e, n, q, st in ks[i0:i1]:
    print("%9.1f us  +%7.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, n))
PY
find $O -name "*kernel_trace.csv" -delete
