#!/bin/bash
# A/B of the key-switch inner product: tools/ab_ks.sh <reps> <variant>[:PT] ...   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for vv in "$@"; do
    v=${vv%%:*}; pt=${vv##*:}; [ "$pt" = "$vv" ] && pt=4
    export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so HP_KS_PT=$pt
    B=256 python $R/tools/bench_stage.py 2>/dev/null | awk -v v=$vv '/ks_inner/{print v, "ks_inner", $3}'
    python $R/bench.py --steps 10 --warmup 2 --roofline-only 2>/dev/null | python -c "import sys,json; print('$vv', 'ckks', round(json.loads(sys.stdin.read())['value']))"
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
