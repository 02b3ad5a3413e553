#!/bin/bash
# Build the library here (hipcc cross-compiles), then run a command on the MI355X box:  tools/gpu.sh [--timeout S] -- 'cmd'
set -e
cd "$(dirname "$0")/.."
python -c 'import __graft_entry__ as g; g.build()' | tail -1
exec /usr/local/graft/bin/gpurun "$@"
