#!/bin/bash
# hom-mult/s (and the digit-spread roofline fraction seen by the library's events) of variants x sub-batch sizes with two
# software-pipelined streams: tools/ab/ab_chunks.sh <reps> "<variants>" "<chunks>"      (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; VARS=$2; CHUNKS=$3
for i in $(seq $REPS); do
for v in $VARS; do for c in $CHUNKS; do
  if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
  if [ "$c" = 0 ]; then S=1; else S=2; fi
  HP_MULT_STREAMS=$S HP_MULT_CHUNK=$c python $R/bench.py --steps 10 --warmup 2 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v chunk=$c', round(r['value']), round(r['roofline']['frac'],3))"
done; done; done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3"/"$4} END{for(k in a) print k":"a[k]}' | sort
