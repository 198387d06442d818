#!/bin/bash
# Round profile: kernel-trace stats for the bench workloads + PMC passes for the NTT shapes.
# usage (GPU box): tools/prof_round.sh <tag>      outputs under gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
cd /tmp && export TMPDIR=/tmp
for wl in ckks ntt intt ntt15 intt15 bgv rotate mul encdec; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_$wl -o p -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --roofline-only > $R/gpurun_out/${TAG}_kt_$wl.log 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/${TAG}_kt_$wl/p_results.db > $R/gpurun_out/${TAG}_kernel_stats_$wl.txt 2>&1
  grep '^{"metric"' $R/gpurun_out/${TAG}_kt_$wl.log >> $R/gpurun_out/${TAG}_bench_lines_under_rocprof.jsonl
  rm -rf $R/gpurun_out/${TAG}_kt_$wl $R/gpurun_out/${TAG}_kt_$wl.log
done
cd $R
tools/prof_pmc.sh ${TAG}_ntt --workload ntt --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_ntt15 --workload ntt15 --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_ckks --workload ckks --steps 2 --warmup 1 --batch 64
