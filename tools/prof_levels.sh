#!/bin/bash
# per-kernel durations of the default C3 step at parity level B and A on one box: tools/prof_levels.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
cd /tmp && export TMPDIR=/tmp
for lvl in B A; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_$lvl -o p -- python $R/bench.py --roofline-only --parity-level $lvl ${@:2} > $R/gpurun_out/${TAG}_kt_$lvl.log 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/${TAG}_kt_$lvl/p_results.db > $R/gpurun_out/${TAG}_kernel_stats_level$lvl.txt 2>&1
  grep '^{"metric"' $R/gpurun_out/${TAG}_kt_$lvl.log > $R/gpurun_out/${TAG}_line_level$lvl.json
  rm -rf $R/gpurun_out/${TAG}_kt_$lvl
done
