#!/bin/bash
# Build an A/B variant of libhdn_hip.so: one source recompiled with extra -D flags, the other objects reused.
#   tools/build_variant.sh <tag> <source.hip> [-DNAME=VALUE ...]   ->  hdn_amd/libhdn_hip_<tag>.so   (use with HDN_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
python -c 'import __graft_entry__ as g; g.build()' | tail -1
tag=$1; src=$2; shift 2
obj=build/obj/${src%.hip}_v_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c hdn_amd/csrc/$src -o $obj
objs=""
for o in build/obj/*.o; do
  b=$(basename $o .o)
  case "$b" in
    ${src%.hip}) ;;                # replaced by the variant object
    *_v_*) ;;                       # other variants
    *) objs="$objs $o" ;;
  esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs $obj -ldl -o hdn_amd/libhdn_hip_$tag.so
echo "built hdn_amd/libhdn_hip_$tag.so"
