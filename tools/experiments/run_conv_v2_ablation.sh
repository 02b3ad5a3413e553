cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ${VARIANTS:-TWICE}; do echo $v; HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_v2$v.so python tools/experiments/exp_conv3x3_v2_time.py 2>&1 | grep "C="; done
done
