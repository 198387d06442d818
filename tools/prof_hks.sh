#!/bin/bash
# HKS extension: bench lines for alpha = k = 2..5, kernel tables for 2 and 5, per-stage timing.  usage (GPU box): tools/prof_hks.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-r01f}; rm -f $R/gpurun_out/${TAG}_hks_bench_lines.jsonl
for a in 2 3 4 5; do
  python $R/bench.py --workload ckks-hks --hks-alpha $a --hks-k $a --steps 20 --warmup 3 --cpu-procs 0 2>&1 | grep '^{"metric"' >> $R/gpurun_out/${TAG}_hks_bench_lines.jsonl
done
cd /tmp && export TMPDIR=/tmp
for a in 2 5; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hks_kt_$a -o p -- python $R/bench.py --workload ckks-hks --hks-alpha $a --hks-k $a --steps 3 --warmup 1 --roofline-only > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/hks_kt_$a/p_results.db > $R/gpurun_out/${TAG}_hks_kernel_stats_alpha$a.txt 2>&1
  rm -rf $R/gpurun_out/hks_kt_$a
done
cd $R; python tools/bench_hks.py 2>&1 | tail -30 > gpurun_out/${TAG}_hks_stages.txt
cat gpurun_out/${TAG}_hks_bench_lines.jsonl | cut -c1-120
