# kernel durations of the first trunk stage from rocprofv3 (host-side launch cost excluded), the shipped library and ablation variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # tag, lib, both
  O=/tmp/prof_$1; rm -rf $O; mkdir -p $O
  HDN_LIB_PATH=$2 BOTH=$3 timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_stem_time.py > $O/log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $1"
  if [ -z "$f" ]; then tail -3 $O/log; else python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "stem" in r["Name"]:
        print("%-48s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  fi
}
run shipped $R/hdn_amd/libhdn_hip.so 1
for v in ${VARIANTS}; do run $v $R/hdn_amd/libhdn_hip_stem$v.so ""; done
