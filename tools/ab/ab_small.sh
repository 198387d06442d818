#!/bin/bash
# A/B of engine variants on the smaller rings: tools/ab/ab_small.sh <reps> <variant> ...   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so
    for wl in ntt intt bgv; do
      python $R/bench.py --workload $wl --steps 20 --warmup 3 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python -c "import sys,json; print('$v', '$wl', round(json.loads(sys.stdin.read())['value']))"
    done
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
