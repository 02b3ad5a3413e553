# ablation builds of conv3x3s2_v2_kernel (tools/build_variant.sh s2<V> conv3x3s2.hip -DHDN_ABLATION -DS2_EXP_<V>): rocprofv3 kernel times at B = 64
cd $GRAFT_REPO_ROOT
echo shipped; bash tools/experiments/run_conv3x3s2_time.sh 2>&1 | grep s2_v2
for v in ${VARIANTS}; do echo $v; bash tools/experiments/run_conv3x3s2_time.sh HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_s2$v.so 2>&1 | grep s2_v2; done
