cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sim
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tests/tools/sequence_bench.py --frames 60 --similarity > $O/log.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "kernels", len(rows))
for r in rows[:40]:
    print("%-95s calls %6s avg_us %7.2f tot_ms %7.2f %5.1f%%" % (r["Name"][:95], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
tail -1 $O/log.txt | cut -c1-600
