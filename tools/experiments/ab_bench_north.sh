for i in 1 2 3; do
for t in "" pre4000 pre2000; do
  if [ -z "$t" ]; then L=$PWD/hdn_amd/libhdn_hip.so; else L=$PWD/hdn_amd/libhdn_hip_$t.so; fi
  HDN_LIB_PATH=$L python bench.py --no-cpu-baseline --no-full-head --no-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('${t:-base}', 'ms/step', round(d['ms_per_step'], 4), 'north mean', round(r['avg_launch_ms']*1e3, 1), 'min', round(r['min_launch_ms']*1e3, 1), 'median', round(r['median_launch_ms']*1e3, 1), 'in-region mean', round(r['timed_region_launch_ms']['mean']*1e3, 1))
"
done; done
