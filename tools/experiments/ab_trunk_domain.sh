# The fused trunk with its interior in the scaled activation domain (one 2^-8 multiply per trunk) against every kernel in real units (one per staged pair
# per layer): HDN_TRUNK_SCALED_DOMAIN=1 / 0, alternating; bench.py's full head (configs[2] per GPU) and the B = 1 estimator inside the sequence block are what move.
for i in 1 2 3; do for m in 1 0; do
  HDN_TRUNK_SCALED_DOMAIN=$m python bench.py --no-cpu-baseline --no-sequence --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); f = d.get('full_head', {})
print(('scaled domain' if $m else 'real units').ljust(16), 'full head ms/step', round(f.get('ms_per_step', float('nan')), 4), 'frames/s', round(f.get('value', float('nan'))), '| kernels-only step ms', round(d['ms_per_step'], 4))"
done; done
