#!/bin/bash
# A/B of one environment knob on the C3 hom-mult rate and the fused drop launch: tools/ab_env2.sh <reps> <VAR> <value> ...   ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=$1; VAR=$2; shift 2
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python $R/bench.py --steps 10 --warmup 2 --roofline-only 2>/dev/null | python -c "import sys,json; print('$VAR=$v', 'ckks', round(json.loads(sys.stdin.read())['value']))"
    python - <<'PY'
import os,sys
R=os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import torch, params as P
from hehub_amd.engine import Engine
eng=Engine(0); mext=P.C3_MODULI_EXT; n=1<<P.C3_LOGN; L=len(mext)-1; B=256
ct=torch.randint(0,1<<40,(B,2,L,n),dtype=torch.int64,device="cuda")
eng.ckks_rescale(mext[:L],ct); torch.cuda.synchronize()
eng.prof_begin("ntt_drop")
for _ in range(5): eng.ckks_rescale(mext[:L],ct)
l,ms=eng.prof_end(); print("%s=%s"%(os.environ.get("ABVAR"),os.environ.get(os.environ.get("ABVAR"),"-")),"drop",round(ms/5,4))
PY
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
