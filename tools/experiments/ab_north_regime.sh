run() {  # $1 tag, rest: env assignments / bench args
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-full-head --no-breakdown $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']
print('$tag', 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean', round(r['avg_launch_ms']*1e3, 1), 'min', round(r['min_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4))"
}
for i in 1 2 3; do
  EXTRA="" run default A=1
  EXTRA="" run spread0 HDN_LIB_PATH=$PWD/hdn_amd/libhdn_hip_spread0.so
  EXTRA="--north fft2w" run fft2w A=1
  EXTRA="" run blocks896 HDN_NORTH_BLOCKS=896
  EXTRA="" run blocks768 HDN_NORTH_BLOCKS=768
done
