# rocprofv3 kernel stats of the full head at B = 64 (tools/experiments/exp_full_head_graph.py); run on the GPU box through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_full
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_full_head_graph.py > $O/log.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:24]:
    print("%-100s calls %6s avg_us %8.2f tot_ms %8.2f %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
