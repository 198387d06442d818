#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of 32 single C3 hom-mults through hehub's object API on ONE lane: what the launches of
# a batch-1 call cost with the split transforms (hp_ntt_split.hip).   gpurun -- bash tools/prof_split.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf /tmp/pp
(cd /tmp && HEHUB_AMD_DEFER=0 HEHUB_AMD_LANES=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- $GRAFT_REPO_ROOT/examples/independent_mults 15 10 32 serial 3 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pp -name "*_results.db" | head -1) | cut -c1-150 | head -24
