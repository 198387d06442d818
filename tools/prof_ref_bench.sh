#!/bin/bash
# hehub's own benchmark (bench/benchmarks.cpp: CKKS rotation at N = 2^12 .. 2^15, create_params' modulus chains) four ways on one box:
#   1. hehub's UNMODIFIED program, hehub alone on one host core            oracle/_ref/ref_bench_cpu   (make -C oracle ref_bench)
#   2. the same program over the binding (hehub's host-memory objects: every call crosses PCIe), without and with the caches
#   3. the benchmark's loop on synthetic words over the own mirror (device-resident objects): examples/rotate_bench, eager on one lane,
#      over four lanes, and recorded (HEHUB_AMD_DEFER=1) -- with hehub's digests from oracle/_ref/ref_rotbench_cpu beside them
#   tools/prof_ref_bench.sh <tag>  -> gpurun_out/<tag>_ref_benchmark.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-refbench}
cd $R; mkdir -p gpurun_out
{
  if [ -x oracle/_ref/ref_bench_cpu ]; then
    echo "== hehub's program, hehub alone (one core): oracle/_ref/ref_bench_cpu"; oracle/_ref/ref_bench_cpu 2>&1 | grep -E "CKKS rotation|ns/op|ms/op"
    echo "== hehub's program over the binding: oracle/_ref/ref_bench_amd"; oracle/_ref/ref_bench_amd 2>&1 | grep -E "CKKS rotation|ns/op|ms/op|hehub_amd"
    echo "== ... HEHUB_AMD_KEY_CACHE=4 HEHUB_AMD_CT_CACHE=64"; HEHUB_AMD_KEY_CACHE=4 HEHUB_AMD_CT_CACHE=64 oracle/_ref/ref_bench_amd 2>&1 | grep -E "CKKS rotation|ns/op|ms/op|hehub_amd"
  fi
  [ -x oracle/_ref/ref_rotbench_cpu ] && { echo "== the benchmark's loop on synthetic words, hehub alone (one core): oracle/_ref/ref_rotbench_cpu 5"; oracle/_ref/ref_rotbench_cpu 5; }
  echo "== own mirror, one lane: HEHUB_AMD_LANES=1 examples/rotate_bench 200"; HEHUB_AMD_DEFER=0 HEHUB_AMD_LANES=1 examples/rotate_bench 200
  echo "== own mirror, default lanes: examples/rotate_bench 200"; HEHUB_AMD_DEFER=0 examples/rotate_bench 200
  echo "== own mirror, recorded: HEHUB_AMD_DEFER=1 examples/rotate_bench 200"; HEHUB_AMD_DEFER=1 examples/rotate_bench 200
  echo "== own mirror, parity level A: HP_PARITY_LEVEL=A examples/rotate_bench 200"; HEHUB_AMD_DEFER=0 HP_PARITY_LEVEL=A examples/rotate_bench 200
} > gpurun_out/${TAG}_ref_benchmark.txt 2>&1
cat gpurun_out/${TAG}_ref_benchmark.txt
