# rocprofv3 kernel times of the B = 1 first stage (vector pipe / matrix cores) and PreShareFeature, every launch cold (256 MB zeroed in between)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # tag, env...
  O=/tmp/prof_$1; rm -rf $O; mkdir -p $O; tag=$1; shift
  env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python $R/tools/experiments/exp_b1_small_kernels.py > $O/log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  if [ -z "$f" ]; then tail -3 $O/log; else python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "stem" in r["Name"] or "share_feature" in r["Name"]:
        print("%-60s calls %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  fi
}
for B in ${BATCHES:-1}; do
run B${B}_vector_pipe B=$B HDN_STEM_MFMA_MIN_BATCH=99
run B${B}_shipped B=$B
done
for v in ${VARIANTS}; do run sf_$v B=1 HDN_LIB_PATH=$R/hdn_amd/libhdn_hip_sf$v.so; done
