cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/prof_s2; rm -rf $O; mkdir -p $O
env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_conv3x3s2_time.py > $O/log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then tail -3 $O/log; else python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv3x3" in r["Name"]:
        n = r["Name"].replace("void hdn::cv::", "").replace("void hdn::cvs::", "").replace("hdn::cvs::", "").replace("hdn::cv::", "")
        print("%-78s calls %4s  avg %7.2f us  min %7.2f" % (n[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
fi
