cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python tools/experiments/exp_conv3x3_time.py
for v in NOLOOP NOEPI NOMFMA; do HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_ab$v.so python tools/experiments/exp_conv3x3_time.py; done
done
