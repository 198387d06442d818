#!/bin/bash
# one rocprofv3 counter pass (own run, --kernel-trace only): tools/pmc_once.sh <tag> "<COUNTER ...>" <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; SET=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $SET --kernel-trace -d $R/gpurun_out/pmc1_$TAG -o p -- python $R/bench.py "$@" --roofline-only > $R/gpurun_out/pmc1_$TAG.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/pmc1_$TAG/p_results.db > $R/gpurun_out/pmc1_${TAG}_summary.txt 2>&1
rm -rf $R/gpurun_out/pmc1_$TAG
grep -E "k_ntt|k_ks|k_tensor" $R/gpurun_out/pmc1_${TAG}_summary.txt | cut -c1-160
