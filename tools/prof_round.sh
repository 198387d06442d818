#!/bin/bash
# Round profile (GPU box): tools/prof_round.sh <tag>      outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/
#   1. the DEFAULT command twice: as the driver runs it (-> <tag>_bench_default.json) and under rocprofv3 --kernel-trace --stats
#      with --roofline-only (k_ntt_fwd<15> then appears in its digit-spread launches only, as the roofline object counts it)
#   2. the full default command under rocprofv3 (its "ntt" / "coeffwise" legs: k_ntt_fwd / k_ntt_inv at N = 4096..32768, k_poly_binary)
#   3. every other workload under rocprofv3 --kernel-trace --stats
#   4. PMC passes (own runs, --kernel-trace only) for the transform shapes and the C3 pipeline
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
cd $R && python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
cd /tmp && export TMPDIR=/tmp
kt() {   # kt <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt_$name -o p -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_kt_$name.log 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/${TAG}_kt_$name/p_results.db > $R/gpurun_out/${TAG}_kernel_stats_$name.txt 2>&1
  grep '^{"metric"' $R/gpurun_out/${TAG}_kt_$name.log >> $R/gpurun_out/${TAG}_bench_lines_under_rocprof.jsonl
  rm -rf $R/gpurun_out/${TAG}_kt_$name $R/gpurun_out/${TAG}_kt_$name.log
}
kt default --roofline-only
kt default_full --cpu-seconds 1 --cpu-procs 0
kt level_a --roofline-only --parity-level A          # the C3 step at the opt-in parity level A (FP64 residue transforms)
for wl in ntt intt ntt15 intt15 bgv rotate mul add encdec; do kt $wl --workload $wl --steps 3 --warmup 1 --roofline-only; done
cd $R
tools/prof_pmc.sh ${TAG}_ntt --workload ntt --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_intt --workload intt --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_ntt15 --workload ntt15 --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_intt15 --workload intt15 --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_ntt12 --workload ntt15 --logn 12 --batch 3724 --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_intt12 --workload intt15 --logn 12 --batch 3724 --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_mul --workload mul --steps 3 --warmup 1
tools/prof_pmc.sh ${TAG}_ckks --workload ckks --steps 2 --warmup 1 --batch 64
tools/prof_pmc.sh ${TAG}_bgv --workload bgv --steps 2 --warmup 1 --batch 128
tools/prof_pmc.sh ${TAG}_ckks_a --workload ckks --steps 2 --warmup 1 --batch 64 --parity-level A
tools/prof_pmc.sh ${TAG}_bgv_a --workload bgv --steps 2 --warmup 1 --batch 128 --parity-level A
# device residency behind hehub's object API: the same program as hehub on the CPU, over the binding, and over the own mirror
tools/prof_resident.sh > gpurun_out/${TAG}_resident_chain.txt 2>&1
# round 5: the HBM bytes of a whole step (both pipelines, both levels), hehub's object API at batch 1 and in all its modes, by-N at both
# levels, the hybrid key switch at both levels, the GPU test tier on the same tree
tools/prof_step_traffic.sh ${TAG}
tools/prof_object_api.sh ${TAG} > gpurun_out/${TAG}_objapi.txt 2>&1
for s in "15 10 256 all 3 8 8 6" "13 6 512 all 3 8 8 6"; do for d in 0 1; do echo "== HEHUB_AMD_DEFER=$d independent_mults $s"; HEHUB_AMD_DEFER=$d examples/independent_mults $s; done; done > gpurun_out/${TAG}_independent_mults.txt 2>&1
tools/prof_matvec.sh ${TAG} > /dev/null 2>&1      # hehub's circuit-level caller: the diagonal loop of matrix_vector_mul_short, every mode + CPU
tools/by_n_levels.sh > gpurun_out/${TAG}_by_n_levels.txt 2>&1
for lv in B A; do python bench.py --workload ckks-hks --parity-level $lv --no-cpu-baseline 2>/dev/null | python $R/tools/benchline.py | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ckks-hks level $lv', round(d['value']), 'hom-mult/s', round(d['ms_per_step'], 3), 'ms per step')"; done > gpurun_out/${TAG}_hks_levels.txt 2>&1
python tools/bench_latency.py > gpurun_out/${TAG}_latency.txt 2>&1
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
