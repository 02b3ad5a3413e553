# The head stream's fork behind the 31x31 launch: wait on the launch's own stop event (no marker on the main stream) against torch's wait_stream.
for i in 1 2 3; do for m in "event step" "stream step" "event fence"; do
  set -- $m
  HDN_BENCH_FORK=$1 HDN_BENCH_JOIN=$2 python bench.py --no-cpu-baseline --no-full-head --no-sequence --no-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('fork on $1, join per $2'.ljust(34), 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean us', round(r['avg_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4))"
done; done
