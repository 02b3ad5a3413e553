# What do the HIP-event brackets around the 31x31 launch cost the timed step?  bench.py with and without them (HDN_BENCH_NO_BRACKETS=1), alternating.
for i in 1 2 3; do for m in "launch 0 1" "launch 0 4" "launch 0 1000" "launch 1 1"; do
  set -- $m; nb=$2
  HDN_BENCH_BRACKETS=$1 HDN_BENCH_NO_BRACKETS=$nb HDN_BENCH_BRACKET_EVERY=$3 python bench.py --no-cpu-baseline --no-full-head --no-sequence --no-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print(('$1-events every $3' if $nb == 0 else 'no brackets in the timed steps').ljust(32), 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean us', round(r['avg_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4))"
done; done
