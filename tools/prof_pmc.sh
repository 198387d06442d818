#!/bin/bash
# PMC passes for one bench workload; counters are collected in their own runs with --kernel-trace only.
# usage (on the GPU box): tools/prof_pmc.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python $R/bench.py "$@" --roofline-only > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python $R/tools/rocpd_summary.py $R/gpurun_out/pmc_${TAG}_*/p_results.db > $R/gpurun_out/pmc_${TAG}_summary.txt 2>&1
rm -rf $R/gpurun_out/pmc_${TAG}_[0-9]* 
