#!/bin/bash
# launch time of the plain forward transform at N = 32768 against the number of rounds (256 limbs each): what a launch costs beyond
# rounds x steady-state time.  tools/ab/sweep_small_launches.sh [variant ...]   ("main" = hehub_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for v in "${@:-main}"; do
  if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
  for B in 93 186 256 419 512; do
    python $R/bench.py --workload ntt15 --batch $B --steps 20 --warmup 3 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$v limbs', $B*11, 'rounds', round($B*11/256,2), 'launch_us', round(ro['avg_launch_ms']*1e3,1), 'us_per_round', round(ro['avg_launch_ms']*1e3/($B*11/256),1))"
  done
done
