#!/bin/bash
# build here (hipcc cross-compiles), then run a command on the GPU box: tools/gpu.sh <timeout s> '<command>'
# (a stale in-tree .so would travel to the box and be measured instead of the sources next to it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R && python -m hehub_amd.build > /dev/null
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
