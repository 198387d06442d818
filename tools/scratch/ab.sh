for v in "" sd2 sd8 sd16; do
  if [ -n "$v" ]; then export HEHUB_AMD_LIB=$PWD/hehub_amd/lib_variants/libhehub_amd_$v.so; else unset HEHUB_AMD_LIB; fi
  for w in ntt15 ntt intt15 intt; do
    python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$v]', '$w', round(d['roofline']['avg_launch_ms'],4), 'ms', round(d['roofline']['achieved']), 'GB/s')"
  done
done
