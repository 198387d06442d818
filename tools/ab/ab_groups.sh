#!/bin/bash
# digit-spread / fused-drop item numbering by groups of G moduli (GPU box): tools/ab/ab_groups.sh "<spread Gs>" "<drop Gs>" [reps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SG=${1:-"2 4"}; DG=${2:-""}; REPS=${3:-3}
for i in $(seq $REPS); do
  for g in $SG; do echo "spread$g $(HP_SPREAD_GROUP=$g python $R/tools/bench_families.py $FAM_ARGS 2>/dev/null)"; done
  for g in $DG; do echo "drop$g $(HP_DROP_GROUP=$g python $R/tools/bench_families.py $FAM_ARGS 2>/dev/null)"; done
done | sort | awk '{k=$1; for(i=2;i<=NF;i++){split($i,a,"="); s[k" "a[1]]+=a[2]; n[k" "a[1]]++}} END{for(x in s) print x, s[x]/n[x]}' | sort | awk '{k=$1; o[k]=o[k]" "$2"="sprintf("%.3f",$3)} END{for(k in o) print k":"o[k]}' | sort
