#!/bin/bash
# build here (hipcc cross-compiles), then run a command on the GPU box: tools/gpu.sh <timeout s> '<command>'
# (a stale in-tree .so or test binary would travel to the box and be measured instead of the sources next to it: the engine, the host
# layer, the example programs and the C++ test binary are all brought up to date first)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R && python -m hehub_amd.build > /dev/null
python - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from hehub_amd.build import build_example
for name in ("independent_mults", "resident_chain", "random_program", "diag_matvec", "rotate_bench"):
    build_example(name)
import test_host_api
test_host_api.build_binary()
PY
# programs compiled against hehub's own headers (oracle/_ref/: prebuilt here, where the reference tree is) follow their sources too
[ -d /root/reference/src ] && make -s -C $R/oracle ref_indep ref_chain ref_randprog ref_randprog_amd ref_matvec ref_rotbench ref_bench ref_tests ref_e2e
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
