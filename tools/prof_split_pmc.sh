#!/bin/bash
# counters of the split-limb transforms at batch 1 (single C3 hom-mults through hehub's object API on one lane):
#   tools/prof_split_pmc.sh <tag>   -> gpurun_out/<tag>_split_pmc_summary.txt   (each counter set in its own run, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-split}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS"; do
  i=$((i+1))
  HEHUB_AMD_DEFER=0 HEHUB_AMD_LANES=1 rocprofv3 --pmc $SET --kernel-trace -d $R/gpurun_out/pmcs_${TAG}_$i -o p -- $R/examples/independent_mults 15 10 8 serial 2 > $R/gpurun_out/pmcs_${TAG}_$i.log 2>&1
done
python $R/tools/rocpd_summary.py $R/gpurun_out/pmcs_${TAG}_*/p_results.db > $R/gpurun_out/${TAG}_split_pmc_summary.txt 2>&1
rm -rf $R/gpurun_out/pmcs_${TAG}_[0-9]*
grep -E "k_ntt_split|k_ks_inner|COUNTER|counter" $R/gpurun_out/${TAG}_split_pmc_summary.txt | cut -c1-220 | head -80
