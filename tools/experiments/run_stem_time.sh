# kernel durations of the first trunk stage from rocprofv3 (host-side launch cost excluded): vector-pipe kernel, matrix-core kernel at 16 / 8 rows per workgroup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # tag, env...
  O=/tmp/prof_$1; rm -rf $O; mkdir -p $O; tag=$1; shift
  env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_stem_time.py > $O/log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  if [ -z "$f" ]; then tail -3 $O/log; else python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "stem" in r["Name"]:
        print("%-60s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  fi
}
for B in ${BATCHES:-64}; do
run "B${B}_rows16" B=$B BOTH=1 HDN_STEM_ROWS=16
run "B${B}_rows8" B=$B HDN_STEM_ROWS=8
done
