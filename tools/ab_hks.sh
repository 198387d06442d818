R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
 for v in base:4 mac:4 mac:2; do
  lib=${v%%:*}; pt=${v##*:}
  export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$lib.so HP_HKS_PT=$pt
  for a in 2 5; do
   python $R/bench.py --workload ckks-hks --hks-alpha $a --hks-k $a --steps 10 --warmup 3 --cpu-procs 0 2>/dev/null | python -c "import sys,json; print('$v', 'alpha$a', round(json.loads(sys.stdin.read())['value']))"
  done
 done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort
