#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of the batched C3 hom-mult at a MID batch (default 8 and 16) through hehub's object
# API (amd::mult_rescale): which launches of the step lose to workgroup quantisation between the split path (<= 128 limbs) and a full
# chip (>= 2048 limbs).   gpurun -- bash tools/prof_midbatch.sh [B ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for B in ${@:-8 16}; do
  rm -rf /tmp/pp
  (cd /tmp && HEHUB_AMD_LANES=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- $GRAFT_REPO_ROOT/examples/independent_mults 15 10 $B batch 12 > /tmp/out.txt 2>&1)
  echo "== batch $B"; grep "^batch " /tmp/out.txt
  python tools/rocpd_summary.py $(find /tmp/pp -name "*_results.db" | head -1) | cut -c1-150 | head -20
done
