# the B = 1 estimator under a graph with ablation builds of conv3x3_kernel (chained form): where do a chained convolution's 6.7 - 8.1 us go?
cd $GRAFT_REPO_ROOT
for v in "" ${VARIANTS}; do
  if [ -z "$v" ]; then L=$PWD/hdn_amd/libhdn_hip.so; else L=$PWD/hdn_amd/libhdn_hip_cv$v.so; fi
  echo "== ${v:-shipped}"
  HDN_LIB_PATH=$L timeout 120 python tools/experiments/exp_homo_b1_profile.py 2>&1 | grep "graph replay\|conv3x3_kernel" | cut -c1-60,100-130 | head -7
done
