#!/bin/bash
# packed digit rows per ring degree (GPU box): interleaved HP_PACK48_MIN_LOGN=15 (N = 32768 only, the round-2 setting until the
# inner product got buffer addressing) vs 11 (every tiled size, the default)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for i in 1 2 3; do for m in 15 11; do
  echo "bgv8192 min$m $(HP_PACK48_MIN_LOGN=$m python $R/tools/bench_families.py --workload bgv)"
  for l in 11 12 13 14; do echo "ckks_logn$l min$m $(HP_PACK48_MIN_LOGN=$m python $R/tools/bench_families.py --logn $l)"; done
done; done 2>/dev/null | sort | awk '{k=$1" "$2; for(i=3;i<=NF;i++){split($i,a,"="); s[k" "a[1]]+=a[2]; n[k" "a[1]]++}} END{for(x in s) print x, s[x]/n[x]}' | sort | awk '{k=$1" "$2; o[k]=o[k]" "$3"="sprintf("%.3f",$4)} END{for(k in o) print k":"o[k]}' | sort
