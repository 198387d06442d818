#!/bin/bash
# round 6, first GPU call: the full GPU tier with the compact bench line; the default bench command as the driver runs it; the host-layer
# tests with deferred mode forced on through the environment (what a flipped default would have to pass)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06a_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r06a_pytest_gpu.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a_bench_default.txt 2> gpurun_out/r06a_bench_default.err ) 2>&1 | grep real
tail -1 gpurun_out/r06a_bench_default.txt | wc -c
tail -1 gpurun_out/r06a_bench_default.txt | head -c 1200; echo
cp bench_sections.json gpurun_out/r06a_bench_sections.json 2>/dev/null
HEHUB_AMD_DEFER=1 python -m pytest tests/test_host_api.py tests/test_object_api.py tests/test_random_program.py tests/test_matvec.py tests/test_rotate_bench.py tests/test_host_residency.py tests/test_extensions.py -m gpu -q > gpurun_out/r06a_pytest_defer.txt 2>&1
tail -15 gpurun_out/r06a_pytest_defer.txt
