# Does giving the bandwidth-bound correlation launches priority over the VALU-bound head stream shorten the step?
#   variant "prio": s_setprio 3 at the top of xcorr_prod29_kernel / xcorr_circ13f_kernel (tools/build_variant.sh prio xcorr.hip -DHDN_ABLATION -DXC_EXP_PRIO=3)
#   HDN_BENCH_HEAD_PRIORITY: queue priority of the head stream (torch.cuda.Stream(priority=...); the correlation stream is the default stream)
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() {
  python bench.py --no-cpu-baseline --no-full-head --no-sequence 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('$1', 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean us', round(r['avg_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4))"
}
for i in 1 2 3; do
  run default
  HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_prio.so run setprio3
  HDN_BENCH_HEAD_PRIORITY=0 run head_prio_0
  HDN_BENCH_HEAD_PRIORITY=-1 run head_prio_-1
  HDN_BENCH_HEAD_PRIORITY=1 run head_prio_1 2>/dev/null
  HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_prio.so HDN_BENCH_HEAD_PRIORITY=0 run setprio3+head_prio_0
done
