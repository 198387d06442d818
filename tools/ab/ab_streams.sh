#!/bin/bash
# hom-mult/s of engine variants x HP_MULT_STREAMS settings, interleaved: tools/ab/ab_streams.sh <reps> <variant> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    for s in 1 2; do
      if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
      HP_MULT_STREAMS=$s python $R/bench.py --steps 10 --warmup 2 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; print('$v streams=$s', round(json.loads(sys.stdin.read())['value']))"
    done
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
