cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # tag env...
  O=/tmp/prof_sf_$1; rm -rf $O; mkdir -p $O; tag=$1; shift
  env "$@" CHECK=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_sf_mc.py > $O/log 2>&1
  echo "== $tag"; grep "max |y" $O/log | head -2
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "share_feature" in r["Name"]:
        print("   %-60s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
run rows HDN_SF_MC=0
for n in ${STRIPS:-8}; do run rmc_strip$n HDN_SF_MC=1 HDN_SF_RMC_STRIP=$n; done
