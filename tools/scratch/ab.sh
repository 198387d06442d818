for b in 23 46 69 92 184 256; do
  python bench.py --workload ntt15 --batch $b --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); W=$b*11; print('batch=$b W=%d'%W, 'waves of WGs=%.2f'%(W/256), round(d['roofline']['avg_launch_ms']*1000,1), 'us', round(d['roofline']['achieved']), 'GB/s')"
done
