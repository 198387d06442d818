for cfg in "1 0" "8 2500" "8 5000" "8 10000" "16 2500" "16 5000" "32 2500"; do set -- $cfg
  export HP_STAGGER_PHASES=$1 HP_STAGGER_TICKS=$2
  for w in ntt15 ntt; do
    python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ph=$1 tk=$2', '$w', round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['achieved']), 'GB/s')"
  done
done
