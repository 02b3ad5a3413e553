# A/B: the 31x31 kernel with the compiler's inter-statement `s_nop 0`s stripped from its assembly (hdn_amd/libhdn_hip_nonop.so, built by hand:
# profiles/round5_north.txt) against the shipping build; parity first, then bench lines alternating
cd $GRAFT_REPO_ROOT
HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_nonop.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "xcorr" 2>&1 | tail -2
for i in 1 2 3; do
for lib in "" nonop; do
  if [ -n "$lib" ]; then export HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_$lib.so; else unset HDN_LIB_PATH; fi
  python bench.py --no-cpu-baseline --no-full-head --no-sequence --no-breakdown 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('${lib:-shipping}', 'value %.1f k  step %.4f ms  north in-step %.1f us (min %.1f) frac %.3f  sustained %.1f us' % (d['value']/1e3, d['ms_per_step'], r['avg_launch_ms']*1e3, r['min_launch_ms']*1e3, r['frac'], r['sustained_launch_ms']*1e3))"
done; done
