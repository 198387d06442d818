#!/bin/bash
# HBM bytes of ONE WHOLE STEP of the pipelines at both parity levels: rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE, each in its own
# run with --kernel-trace only) of the default command's timed region (bench.py --roofline-only: default batch, 2 timed + 1 warm-up step).
#   gpurun -- tools/prof_step_traffic.sh <tag>   ->  gpurun_out/<tag>_pmc_step_{ckks,bgv}{,_a}_summary.txt
#   python tools/step_traffic_from_pmc.py <tag> gpurun_out   ->  profiles/traffic.json "step_<workload>_<level>" (read by bench.py: step.step_traffic)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-step}
cd /tmp && export TMPDIR=/tmp
for WL in ckks bgv; do
  for LV in B A; do
    SUF=""; [ $LV = A ] && SUF="_a"
    OUT=$R/gpurun_out/${TAG}_pmc_step_${WL}${SUF}
    i=0
    for SET in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      rocprofv3 --pmc $SET --kernel-trace -d ${OUT}_$i -o p -- python $R/bench.py --workload $WL --parity-level $LV --steps 2 --warmup 1 --roofline-only > ${OUT}_$i.log 2>&1
    done
    python $R/tools/rocpd_summary.py ${OUT}_*/p_results.db > ${OUT}_summary.txt 2>&1
    rm -rf ${OUT}_[0-9]*
    grep -cE "FETCH_SIZE|WRITE_SIZE" ${OUT}_summary.txt
  done
done
