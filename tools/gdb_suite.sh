#!/bin/bash
# the GPU test tier under rocgdb until a run dies (intermittent aborts): tools/gdb_suite.sh <max runs> [pytest args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
cd $R
for i in $(seq $N); do
  rocgdb -batch -ex "set pagination off" -ex "handle SIGPIPE nostop noprint pass" -ex run -ex "bt 40" -ex "info threads" -ex "thread apply all bt 25" \
    --args python -m pytest tests -m gpu -x -q "$@" > gpurun_out/gdb_suite_$i.txt 2>&1
  if grep -q "passed" gpurun_out/gdb_suite_$i.txt && ! grep -q "SIGABRT\|SIGSEGV\|Aborted" gpurun_out/gdb_suite_$i.txt; then
    echo "run $i: $(grep -h ' passed' gpurun_out/gdb_suite_$i.txt | tail -1)"; rm gpurun_out/gdb_suite_$i.txt
  else
    echo "run $i DIED"; grep -n "SIGABRT\|SIGSEGV" -A45 gpurun_out/gdb_suite_$i.txt | head -120; break
  fi
done
