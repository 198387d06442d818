#!/bin/bash
# A/B of one environment knob with interleaved repetitions (GPU box): tools/ab/ab_env.sh <reps> <VAR> <value> <value> ...   ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; VAR=$2; shift 2
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python $R/tools/bench_stage.py 2>/dev/null | awk -v v=$v '/spread/{print "'$VAR'="v, "spread", $6}'
    python $R/bench.py --steps 10 --warmup 2 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python -c "import sys,json; print('$VAR=$v', 'ckks', round(json.loads(sys.stdin.read())['value']))"
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
