R=$GRAFT_REPO_ROOT
for wl in ntt15 intt15; do for B in 256 512 1024 2327; do
python $R/bench.py --workload $wl --batch $B --steps 10 --warmup 3 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); ro=r['roofline']; print('$wl B=$B limbs', $B*11, 'ms', round(ro['avg_launch_ms'],4), 'us/round', round(ro['avg_launch_ms']*1e3/($B*11/256),2), 'frac', round(ro['frac'],3))"
done; done
