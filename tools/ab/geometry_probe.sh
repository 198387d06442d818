#!/bin/bash
# "Two limbs in flight per CU" (VERDICT r04 item 2): the geometry exists at N = 16384 (512-thread workgroups, 73 KB of LDS: two per
# CU, their load / exchange / tail phases overlap each other's arithmetic).  How much does it buy?  The digit-spread launch of the C3
# chain at N = 16384 with the bytes of the N = 32768 launch (batch 512 against 256), both levels: roofline.frac side by side.
#   gpurun -- tools/ab/geometry_probe.sh > gpurun_out/<tag>_geometry_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lvl in B A; do
  for shape in "15 256" "14 512" "13 1024"; do
    set -- $shape
    python bench.py --workload ckks --logn $1 --batch $2 --parity-level $lvl --no-cpu-baseline --no-verify --steps 10 --warmup 2 2>/dev/null | python $R/tools/benchline.py | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('level $lvl N=%6d batch %4d: spread launch %.3f ms, frac %.3f, %.0f hom-mult/s (x N/32768: %.0f), step %.3f ms' % (1 << $1, $2, r['avg_launch_ms'], r['frac'], d['value'], d['value'] * (1 << $1) / 32768, d['ms_per_step']))"
  done
done
