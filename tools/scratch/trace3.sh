for v in trace_cls0 trace_bglobal; do
  sed "s/libhehub_amd_trace.so/libhehub_amd_$v.so/" tools/scratch/trace2.py > /tmp/t3.py
  echo "== $v"; python /tmp/t3.py 2>&1 | grep -v amdgpu | grep "W= 2816"
done
