#!/bin/bash
# hehub's object API at batch 1: where the time of ONE ckks::mult + rescale_inplace goes (host enqueue vs kernels vs gaps).
#   gpurun -- tools/prof_object_api.sh <tag>    -> gpurun_out/<tag>_objapi_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-objapi}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
HEHUB_AMD_DEFER=0 examples/independent_mults 15 10 32 serial 3 > gpurun_out/${TAG}_objapi_serial.txt 2>&1
HEHUB_AMD_DEFER=0 examples/independent_mults 15 10 32 chains 3 8 8 6 > gpurun_out/${TAG}_objapi_chains.txt 2>&1
cat gpurun_out/${TAG}_objapi_serial.txt gpurun_out/${TAG}_objapi_chains.txt
rm -rf /tmp/prof_objapi
(cd /tmp && HEHUB_AMD_DEFER=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_objapi -o t -- $R/examples/independent_mults 15 10 32 serial 3 > /dev/null 2>&1)
python $R/tools/rocpd_summary.py $(find /tmp/prof_objapi -name "*_results.db" | head -1) > gpurun_out/${TAG}_objapi_kernel_stats.txt 2>&1
head -40 gpurun_out/${TAG}_objapi_kernel_stats.txt
