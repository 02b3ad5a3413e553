cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mc in 0 1; do
  O=/tmp/prof_sf$mc; rm -rf $O; mkdir -p $O
  HDN_SF_MC=$mc CHECK=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_sf_mc.py > $O/log 2>&1
  grep "B=" $O/log
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "share_feature" in r["Name"]:
        print("   %-60s calls %4s  avg %7.2f us  min %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
