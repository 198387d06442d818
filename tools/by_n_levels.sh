#!/bin/bash
# hom-mult/s at N = 4096 .. 32768 (C3's moduli, batch 256) and rotations/s at both parity levels, every output verified:
#   tools/by_n_levels.sh > gpurun_out/<tag>_by_n_levels.txt        (last column: A_step fraction of HBM peak)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lvl in B A; do
  for logn in 12 13 14 15; do
    python bench.py --workload ckks --logn $logn --parity-level $lvl --no-cpu-baseline 2>/dev/null | python $R/tools/benchline.py | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ckks level $lvl N=%6d' % (1 << $logn), round(d['value']), d.get('verified'), round(d.get('pipeline_roofline', {}).get('frac_of_hbm_peak', 0), 3))"
  done
  python bench.py --workload rotate --parity-level $lvl --no-cpu-baseline 2>/dev/null | python $R/tools/benchline.py | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotate level $lvl', round(d['value']), d.get('verified'))"
done
