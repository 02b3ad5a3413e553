cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_full_head
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/bench.py --workload full --steps 30 --warmup 10 --no-full-head > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:28]:
    n=r["Name"].replace("void hdn::cv::","").replace("hdn::cv::","")[:95]
    print("%-95s calls %5s  avg %8.1f us  total %8.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
cp $f $R/gpurun_out/round5_kernel_stats_full_head.csv
find $O -name "*.csv" -size +20M -delete
