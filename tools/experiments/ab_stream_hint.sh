# bench.py step with / without the streaming hints (hdn_amd/libhdn_hip_nohint.so: xcorr.hip and xcorr_fft.hip built with -DHDN_STREAM_HINT=0)
for i in 1 2 3; do for t in "" nohint; do
  if [ -z "$t" ]; then L=$PWD/hdn_amd/libhdn_hip.so; else L=$PWD/hdn_amd/libhdn_hip_$t.so; fi
  HDN_LIB_PATH=$L python bench.py --no-cpu-baseline --no-full-head 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); r = d['roofline']; k = d['kernels']
print('${t:-hints}', 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| north mean', round(r['avg_launch_ms']*1e3, 1), 'min', round(r['min_launch_ms']*1e3, 1), 'frac', round(r['frac'], 4), '| solo us: north', round(k['xcorr_31x31_61x61']['ms']*1e3,1), 'prod29', round(k['xcorr_5x5_29x29_x6']['ms']*1e3,1), 'circ13', round(k['xcorr_circ_13x13_x6']['ms']*1e3,1))"
done; done
