# The 13x13 circular kernel as a persistent grid with the next group's loads in flight during the compute stages: HDN_CIRC13_WGS = 0 (one group per wave) / N workgroups.
for i in 1 2; do for w in 0 768 911 1366 2048; do
  HDN_CIRC13_WGS=$w python bench.py --no-cpu-baseline --no-full-head --no-sequence 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); k = d['kernels']
print('workgroups $w'.ljust(18), 'ms/step', round(d['ms_per_step'], 4), 'frames/s', round(d['value']), '| alone: circ13 x6', round(k['xcorr_circ_13x13_x6']['ms']*1e3, 1), 'us =', round(k['xcorr_circ_13x13_x6']['algorithmic_GBps']/8000, 3))"
done; done
