cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" 1 2 3; do
  O=/tmp/prof_sfm$v; rm -rf $O; mkdir -p $O
  if [ -z "$v" ]; then L=$R/hdn_amd/libhdn_hip.so; else L=$R/hdn_amd/libhdn_hip_sfm$v.so; fi
  HDN_LIB_PATH=$L HDN_SF_MC=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_sf_mc.py > $O/log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "stop after phase ${v:-none}"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "share_feature" in r["Name"]:
        print("   %-60s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
