for i in 1 2 3; do for m in parallel after-north; do
  python bench.py --no-cpu-baseline --no-full-head --no-breakdown --head-stream $m 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('$m', 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean', round(r['avg_launch_ms']*1e3, 1), 'in-region', round(r['timed_region_launch_ms']['mean']*1e3, 1))"
done; done
