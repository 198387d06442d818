cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for q in 4 8; do
rm -rf /tmp/pc; (cd /tmp && HEHUB_AMD_DEFER=0 GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace -d /tmp/pc -o t -- $GRAFT_REPO_ROOT/examples/independent_mults 15 10 32 chains 3 8 8 6 | tail -3)
python tools/trace/concurrency.py $(find /tmp/pc -name "*_results.db" | head -1) 0.6 | tee gpurun_out/r05a_concurrency_q$q.txt
done
