#!/bin/bash
# interleaved A/B of engine variants per kernel family (GPU box): tools/ab/ab_families.sh <reps> <variant> <variant> ...   ("main" = hehub_amd/lib)
# extra arguments for bench_families.py through FAM_ARGS, e.g. FAM_ARGS="--workload bgv"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    echo "$v $(python $R/tools/bench_families.py $FAM_ARGS 2>/dev/null)"
  done
done | python3 -c "
import sys, collections, statistics
d=collections.defaultdict(lambda: collections.defaultdict(list))
for line in sys.stdin:
    t=line.split()
    for kv in t[1:]:
        k,v=kv.split('='); d[t[0]][k].append(float(v))
for v,fam in d.items():
    print(v, ' '.join(f'{k}={statistics.median(x):.3f}' for k,x in fam.items()), f'(n={len(next(iter(fam.values())))})')
"
