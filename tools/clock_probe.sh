#!/bin/bash
# shader clock / power while the digit-spread transform runs (GPU box): tools/clock_probe.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python bench.py --workload ntt15 --batch 2327 --steps 6000 --warmup 5 --roofline-only > /tmp/probe_bench.json 2>/dev/null &
BP=$!
for i in $(seq 60); do
  L=$(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Socket Graphics|Sensor junction" | sed 's/.*: //' | tr '\n' ' ')
  echo "t=$i $L"
  kill -0 $BP 2>/dev/null || break
  sleep 1
done | awk '{print}' | grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" | head -20
wait $BP
python3 -c "import json; r=json.loads(open('/tmp/probe_bench.json').read().strip().splitlines()[-1]); print('ntt15 steady: frac', round(r['roofline']['frac'],4), 'avg_launch_ms', round(r['roofline']['avg_launch_ms'],4))"
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
