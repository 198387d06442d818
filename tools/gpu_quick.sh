#!/bin/bash
# quick GPU check after a kernel change: transform / pipeline parity tests, then the default bench line -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-quick}
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_full_batch.py tests/test_hks.py tests/test_extensions.py tests/test_sharded.py -m gpu -x -q > gpurun_out/${TAG}_pytest.txt 2>&1
tail -3 gpurun_out/${TAG}_pytest.txt
python bench.py --steps 10 --warmup 2 --cpu-seconds 2 --cpu-procs 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json, sys
sys.path.insert(0, '.')
from benchkit.line import collect
r=collect(open('gpurun_out/${TAG}_bench.json').read())[1]
import os
if os.path.exists('bench_sections.json'): r.update({k: v for k, v in json.load(open('bench_sections.json')).items() if k in r})   # every level of the sections
print('hom-mult/s', round(r['value']), 'verified', r.get('verified'), 'spread ms', round(r['roofline']['avg_launch_ms'],3), 'frac', round(r['roofline']['frac'],3))
for n,e in r['ntt']['by_N'].items():
    print(n, 'fwd', round(e['forward']['frac_of_hbm_peak'],3), 'inv', round(e['inverse']['frac_of_hbm_peak'],3), e.get('verified'))
for n,e in r.get('ckks_by_N',{}).items():
    print('ckks', n, round(e['per_s']), round(e['A_step_frac_of_hbm_peak'],3), e.get('verified'))
PY
