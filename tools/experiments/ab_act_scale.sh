#!/bin/bash
# Does splitting the activations as x 2^-8 (csrc/mfma_split.h, ABI 9) cost time?  A/B of the shipping library against one built with
# -DHDN_ACT_SCALE_LOG2=0 (the five matrix-core sources recompiled, the other objects reused):
#   here:        bash tools/experiments/ab_act_scale.sh build
# (both arms with HDN_TRUNK_SCALED_DOMAIN=0: the per-layer form is what is compared; hdn_amd.trunk's scaled domain assumes the 2^-8 of the shipping build)
#   on the box:  bash tools/experiments/ab_act_scale.sh run     -> full head (configs[2] per GPU share, 64 pairs) and B = 1 estimator times, 3 alternations
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  python -c 'import __graft_entry__ as g; g.build()' | tail -1
  objs=""
  for o in build/obj/*.o; do
    b=$(basename $o .o)
    case "$b" in
      conv3x3|conv3x3s2|trunk_stem_mfma|head_conv|head_tail)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DHDN_ACT_SCALE_LOG2=0 -c hdn_amd/csrc/$b.hip -o build/obj/${b}_v_noscale.o
        objs="$objs build/obj/${b}_v_noscale.o" ;;
      *_v_*) ;;
      *) objs="$objs $o" ;;
    esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -ldl -o hdn_amd/libhdn_hip_noscale.so
  echo "built hdn_amd/libhdn_hip_noscale.so"
  exit 0
fi
for i in 1 2 3; do for t in "" noscale; do
  if [ -z "$t" ]; then L=$PWD/hdn_amd/libhdn_hip.so; else L=$PWD/hdn_amd/libhdn_hip_$t.so; fi
  HDN_TRUNK_SCALED_DOMAIN=0 HDN_LIB_PATH=$L python bench.py --no-cpu-baseline --no-sequence --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); f = d.get('full_head', {})
print('${t:-x 2^-8 (shipping)}'.ljust(22), 'full head ms/step', round(f.get('ms_per_step', float('nan')), 4), 'frames/s', round(f.get('value', float('nan'))), '| kernels-only step ms', round(d['ms_per_step'], 4))"
done; done
