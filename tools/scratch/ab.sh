for v in "" v120 v112; do
  if [ -n "$v" ]; then export HEHUB_AMD_LIB=$PWD/hehub_amd/lib_variants/libhehub_amd_$v.so; else unset HEHUB_AMD_LIB; fi
  for w in ntt15; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$v]', '$w', round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['achieved']), 'GB/s')"
  done
  for cfg in "1 256" "2 128" "2 64"; do set -- $cfg
    HP_MULT_STREAMS=$1 HP_MULT_CHUNK=$2 python bench.py --workload ckks --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v streams=$1 chunk=$2] ckks', round(d['value']), round(d['ms_per_step'],3))"
  done
done
