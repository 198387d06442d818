R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_hks.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2 3 4; do
  for v in 0 2; do
    export HP_SPREAD_GROUP=$v
    python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('group=$v', 'ckks', round(json.loads(sys.stdin.read())['value']))"
    python - <<'PY'
import os,sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]+"/tests")
import torch, params as P
from hehub_amd.engine import Engine
eng=Engine(0); mext=P.C3_MODULI_EXT; n=1<<P.C3_LOGN; L=len(mext)-1; B=256
ct=torch.randint(0,1<<40,(B,2,L,n),dtype=torch.int64,device="cuda")
eng.ckks_rescale(mext[:L],ct); torch.cuda.synchronize()
eng.prof_begin("ntt_drop")
for _ in range(5): eng.ckks_rescale(mext[:L],ct)
l,ms=eng.prof_end(); print("group=%s"%os.environ["HP_SPREAD_GROUP"],"drop",round(ms/5,4))
PY
  done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
