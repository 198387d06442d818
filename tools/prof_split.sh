cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf /tmp/pp
(cd /tmp && HEHUB_AMD_LANES=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- $GRAFT_REPO_ROOT/examples/independent_mults 15 10 32 serial 3 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find /tmp/pp -name "*_results.db" | head -1) | cut -c1-150 | head -24
