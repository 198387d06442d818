#!/bin/bash
# full GPU test tier, the default bench line, PMC passes of the inverse transform: tools/gpu_round_check.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/round_pytest_gpu.txt 2>&1
tail -5 gpurun_out/round_pytest_gpu.txt
python bench.py > gpurun_out/round_bench_default.json 2> gpurun_out/round_bench_default.err
tail -c 1500 gpurun_out/round_bench_default.json; tail -3 gpurun_out/round_bench_default.err
for spec in "intt15:--workload intt15" "intt14:--workload intt" "intt13:--workload intt15 --logn 13" "intt12:--workload intt15 --logn 12"; do
  tag=${spec%%:*}; flags=${spec#*:}
  tools/prof_pmc.sh round_$tag $flags --steps 3 --warmup 1
done
ls gpurun_out | head -40
