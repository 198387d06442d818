#!/bin/bash
# interleaved A/B of the hom-mult step at both parity levels: tools/ab/ab_level_a.sh <reps> <variant> ...   ("main" = hehub_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    python $R/tools/time_levels.py 10 2>/dev/null | awk -v v=$v '/level/{gsub(":","",$4); print v, $1 $2 "_" $4 "=" $5}' | tr '\n' ' ' | awk -v v=$v '{out=v; for(i=2;i<=NF;i+=2) out=out" "$i; print out}'
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
