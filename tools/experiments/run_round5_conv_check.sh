cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or full_head or trunk" 2>&1 | tail -5
python bench.py --workload full --steps 30 --warmup 10 --no-full-head 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full workload', d['value'], d['ms_per_step'])"
python tools/experiments/exp_full_head_streams.py 2>&1 | tail -8
HDN_CV2_SLICE_TARGET=100 python tools/experiments/exp_full_head_streams.py 2>&1 | tail -8
