#!/bin/bash
# the diagonal loop of matrix_vector_mul_short (examples/diag_matvec) at the C3 shape: times of every mode, hehub on the CPU beside it
# (oracle/_ref/ref_matvec_cpu, prebuilt), kernel stats of the deferred and the eager pass: tools/prof_matvec.sh <tag> [width=16]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-matvec}; W=${2:-16}
cd $R; mkdir -p gpurun_out
{
  for m in short full; do
    echo "== examples/diag_matvec 15 10 $W $m 5"; examples/diag_matvec 15 10 $W $m 5
    echo "== HP_PARITY_LEVEL=A examples/diag_matvec 15 10 $W $m 5"; HP_PARITY_LEVEL=A examples/diag_matvec 15 10 $W $m 5
  done
  echo "== MATVEC_HOST_DIAGS=1 (diagonals stay host objects: copied and uploaded in every mult_plain)"; MATVEC_HOST_DIAGS=1 examples/diag_matvec 15 10 $W short 5
  echo "== examples/diag_matvec 13 6 $W short 5"; examples/diag_matvec 13 6 $W short 5
  [ -x oracle/_ref/ref_matvec_cpu ] && { echo "== hehub on the CPU: oracle/_ref/ref_matvec_cpu 15 10 $W short 1"; oracle/_ref/ref_matvec_cpu 15 10 $W short 1; oracle/_ref/ref_matvec_cpu 15 10 $W full 1; }
} > gpurun_out/${TAG}_matvec.txt 2>&1
export TMPDIR=/tmp
for pass in 0 1; do
  rm -rf /tmp/pm; (cd /tmp && MATVEC_PASS=$pass rocprofv3 --kernel-trace --stats -d /tmp/pm -o t -- $R/examples/diag_matvec 15 10 $W short 6 > /dev/null 2>&1)
  python $R/tools/rocpd_summary.py $(find /tmp/pm -name "*_results.db" | head -1) > gpurun_out/${TAG}_matvec_kernel_stats_pass$pass.txt 2>&1
  head -24 gpurun_out/${TAG}_matvec_kernel_stats_pass$pass.txt
done
cat gpurun_out/${TAG}_matvec.txt
