#!/bin/bash
# Ring-degree sweep: tools/ab/sweep_n.sh > gpurun_out/<tag>_sweep_N.txt   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
echo "# bench.py --workload {ntt15,intt15,ckks} --logn {12..15} --steps 10 --warmup 2 on one MI355X (batch 256 polynomials x 11"
echo "# limbs for the transforms; 256 ciphertext pairs, L = 10 moduli + special prime for CKKS mult+relin+rescale)."
echo "# columns: workload, log2 N, GPU value, unit, forward/inverse-NTT roofline GB/s and fraction of 8 TB/s,"
echo "#          A_step pipeline fraction of 8 TB/s, compiled reference on one core of the same host"
for logn in 12 13 14 15; do
  for wl in ntt15 intt15 ckks; do
    python $R/bench.py --workload $wl --logn $logn --steps 10 --warmup 2 2>/dev/null | python $R/tools/benchline.py | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); p=d.get('pipeline_roofline',{}); c=d.get('cpu_baseline',{})
print('$wl', $logn, round(d['value']), d['unit'], 'roof', round(r.get('achieved',0)), round(r.get('frac',0),3), 'pipe', round(p.get('frac_of_hbm_peak',0),3), 'cpu', round(c.get('value',0),1))"
  done
done
