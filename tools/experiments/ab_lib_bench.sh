# bench.py step with the default library vs variants: ab_lib_bench.sh <tag> [<tag> ...]   (hdn_amd/libhdn_hip_<tag>.so from tools/build_variant.sh)
for i in 1 2 3; do for t in "" "$@"; do
  if [ -z "$t" ]; then L=$PWD/hdn_amd/libhdn_hip.so; else L=$PWD/hdn_amd/libhdn_hip_$t.so; fi
  HDN_LIB_PATH=$L python bench.py --no-cpu-baseline --no-full-head --no-sequence 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; k = d['kernels']
print('${t:-default}', 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean', round(r['avg_launch_ms']*1e3, 1), 'min', round(r['min_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4), '| solo us: north', round(k['xcorr_31x31_61x61']['ms']*1e3,1), 'prod29', round(k['xcorr_5x5_29x29_x6']['ms']*1e3,1), 'circ13', round(k['xcorr_circ_13x13_x6']['ms']*1e3,1))"
done; done
