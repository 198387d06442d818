#!/bin/bash
# interleaved A/B of the inverse transform at several sizes (GPU box): VARIANTS="base main" tools/ab/ab_inverse.sh
R=$GRAFT_REPO_ROOT
cd $R; python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_golden.py tests/test_extensions.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2 3; do for v in ${VARIANTS:-base main}; do
if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
for spec in "intt15:--batch:2327" "intt15:--batch:512" "intt:--batch:2560" "intt15:--logn 13 --batch:1860" "intt15:--logn 12 --batch:3724"; do
 wl=${spec%%:*}; rest=${spec#*:}; fl=${rest%%:*}; b=${rest##*:}
 python bench.py --workload $wl $fl $b --steps 8 --warmup 2 --roofline-only 2>/dev/null | python $R/tools/benchline.py | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', '$wl$fl$b'.replace(' ','_'), round(r['roofline']['frac'],4))"
done; done; done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3} END{for(k in a) print k":"a[k]}' | sort -k2
