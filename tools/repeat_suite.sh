#!/bin/bash
# the GPU test tier N times in a row, stopping at the first run that does not pass: tools/repeat_suite.sh <runs> [pytest args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
cd $R
for i in $(seq $N); do
  python -m pytest tests -m gpu -x -q "$@" > gpurun_out/repeat_suite_$i.txt 2>&1
  if tail -1 gpurun_out/repeat_suite_$i.txt | grep -q " passed" ; then echo "run $i: $(tail -1 gpurun_out/repeat_suite_$i.txt)"; rm gpurun_out/repeat_suite_$i.txt
  else echo "run $i FAILED"; grep -v "^  File\|^$" gpurun_out/repeat_suite_$i.txt | tail -15; break; fi
done
