#!/bin/bash
# the full GPU tier (no -x: every failure is listed), then the default bench command as the driver runs it: tools/gpu_tier.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
cd $R; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1 ) 2>&1 | grep real
tail -12 gpurun_out/${TAG}_pytest_gpu.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.txt 2> gpurun_out/${TAG}_bench_default.err ) 2>&1 | grep real
tail -1 gpurun_out/${TAG}_bench_default.txt | wc -c
tail -1 gpurun_out/${TAG}_bench_default.txt | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], json.dumps(r['summary']))"
cp bench_sections.json gpurun_out/${TAG}_bench_sections.json 2>/dev/null
tail -3 gpurun_out/${TAG}_bench_default.err
