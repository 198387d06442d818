#!/bin/bash
# A/B of engine variants with interleaved repetitions: tools/ab/ab.sh <reps> <variant> <variant> ...   (GPU box)
# prints per variant the sorted per-run times of the C3 digit-spread and in-place transform launches and the hom-mult/s
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so
    B=256 python $R/tools/bench_stage.py | awk -v v=$v '/spread/{s=$6} /in-place/{p=$6} END{print v, "spread", s, "inplace", p}'
    python $R/bench.py --steps 10 --warmup 2 --roofline-only | python $R/tools/benchline.py | python -c "import sys,json; print('$v', 'ckks', round(json.loads(sys.stdin.read())['value']))"
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3; if($4!=""){k2=$1" "$4; a[k2]=a[k2]" "$5}} END{for(k in a) print k":"a[k]}' | sort
