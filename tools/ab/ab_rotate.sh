#!/bin/bash
# interleaved A/B of the rotation workload + drop families: tools/ab/ab_rotate.sh <reps> <variant> <variant> ...  ("main" = hehub_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    rot=$(python $R/bench.py --workload rotate --steps 10 --warmup 3 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
    echo "$v rot=$rot $(python $R/tools/bench_families.py 2>/dev/null) $(FAM=1 python $R/tools/bench_families.py --workload bgv 2>/dev/null | sed 's/\([a-z_]*\)=/bgv_\1=/g')"
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
