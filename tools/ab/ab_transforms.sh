#!/bin/bash
# interleaved A/B of the plain transforms at N = 32768 (C3 moduli): forward / inverse, C3-batch-sized (5120 limbs) and steady (25600) launches,
# plus the per-family times of the C3 step.  tools/ab/ab_transforms.sh <reps> <variant> <variant> ...   ("main" = hehub_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    echo "$v $(python $R/tools/bench_transforms.py 2>/dev/null) $(python $R/tools/bench_families.py $FAM_ARGS 2>/dev/null)"
  done
done | python3 -c "
import sys, collections, statistics
d=collections.defaultdict(lambda: collections.defaultdict(list))
for line in sys.stdin:
    t=line.split()
    for kv in t[1:]:
        k,v=kv.split('='); d[t[0]][k].append(float(v))
for v,fam in d.items():
    print(v, ' '.join(f'{k}={statistics.median(x):.4g}' for k,x in fam.items()), f'(n={len(next(iter(fam.values())))})')
"
