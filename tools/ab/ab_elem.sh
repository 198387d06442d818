#!/bin/bash
# interleaved A/B of the coefficient-wise kernels (C3 limb shape, 3 x 512 MiB per launch): tools/ab/ab_elem.sh <reps> <variant> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    m=$(python $R/bench.py --workload mul --steps 30 --warmup 5 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['frac'],4))")
    a=$(python $R/bench.py --workload add --steps 30 --warmup 5 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['frac'],4))")
    echo "$v mul=$m add=$a"
  done
done
