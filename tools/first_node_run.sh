#!/bin/bash
# First run on an 8-GPU MI355X node: everything the multi-GPU rows of DESIGN.md section 6 still owe, in one go.
#   tools/first_node_run.sh [gpus = all visible] [outdir = gpurun_out/node]
# On a one-GPU box it runs with gpus = 1; HP_NODE_TEST_RANKS=8 makes the multi-rank steps share GPU 0 (gloo rendezvous, ranks
# {0,0,...}): every code path, no scaling number.  What each step produces:
#   1 peer_matrix.txt     hp_node_peer_matrix of all devices (1 = direct peer writes over xGMI, 0 = staged through pinned host memory)
#                         + hp_node_placement: the NUMA node / CPUs every rank's worker thread was bound to
#   2 scale.json          bench.py --gpus {1,2,4,8} (C4 = C3 per GPU, batch-sharded, weak scaling): one SCALE-shaped object
#                         {"runs": [line per N], "efficiency": value_N / (N * value_1)} -- expected ~1.0: no data-path collective
#   3 bgv_scale.json      the same for --workload bgv (C5: 4096 pairs over 8 GPUs)
#   4 limb_p2p.json / limb_allgather.json   bench.py --workload ckks-limb --gpus N with both digit-exchange transports (latency mode,
#                         strong scaling; bound (L+1)/ceil((L+1)/N) = 5.5 x at 8 GPUs)
#   5 node_batch.txt      examples/node_batch on N real devices (the C node layer: host-resident and device-resident batches)
#   6 rccl_allgather.txt  RCCL all_gather bandwidth, 2^20 .. 2^28 bytes per rank (the exchange size of the limb-sharded digits is
#                         batch x L x N x 8 / ranks bytes per rank)
#   7 objapi_scale.txt    hehub's OBJECT API over 1 / 2 / 4 / 8 device ranks (HEHUB_AMD_DEVICES; examples/independent_mults at the C3 shape,
#                         B = 256 per device): the unchanged loop of single calls, the batched form, independent chains -- rates and digests
#   8 node_transports.txt the limb-sharded plan of the C node layer under its three transports (peer writes / RCCL / packed copies)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
VIS=$(python -c "import torch; print(torch.cuda.device_count())")
G=${1:-$VIS}
OUT=${2:-gpurun_out/node}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
TEST_RANKS=${HP_NODE_TEST_RANKS:-0}
SIZES=""; for n in 1 2 4 8; do [ $n -le $G ] && SIZES="$SIZES $n"; done
[ $TEST_RANKS -gt 1 ] && SIZES="1 2 $TEST_RANKS" && export HP_BENCH_SHARE_GPU=1
echo "devices visible $VIS, using $G, job sizes:$SIZES $([ $TEST_RANKS -gt 1 ] && echo '(TEST MODE: ranks share GPU 0)')" | tee $OUT/README.txt

# 1 ---- peer matrix + placement -------------------------------------------------------------------------------------------------
python - > $OUT/peer_matrix.txt 2>&1 <<PY
import ctypes as C
from hehub_amd import capi
lib = capi.load()
n = max($G, $TEST_RANKS)
devs = (C.c_int * n)(*[(i if $TEST_RANKS <= 1 else 0) for i in range(n)])
node = C.c_void_p()
assert lib.hp_node_create(devs, n, C.byref(node)) == 0
m = (C.c_int * (n * n))()
lib.hp_node_peer_matrix(node, m)
print("peer matrix (row a can write column b directly):")
for a in range(n):
    print(" ".join(str(m[a * n + b]) for b in range(n)))
for r in range(n):
    numa, cpus = C.c_int(-2), C.c_int(0)
    lib.hp_node_placement(node, r, C.byref(numa), C.byref(cpus))
    buf = C.create_string_buffer(512); nd = C.c_int(-1)
    lib.hp_device_numa(devs[r], C.byref(nd), buf, 512)
    print(f"rank {r}: device {devs[r]} numa_node {numa.value} worker bound to {cpus.value} CPUs (node cpulist '{buf.value.decode()}')")
lib.hp_node_destroy(node)
PY
cat $OUT/peer_matrix.txt

# 2, 3 ---- batch-sharded scaling -------------------------------------------------------------------------------------------------
for WL in ckks bgv; do
  F=$OUT/scale.json; [ $WL = bgv ] && F=$OUT/bgv_scale.json
  : > $OUT/lines_$WL.txt
  for n in $SIZES; do
    python bench.py --gpus $n --workload $WL --no-cpu-baseline --no-rates --no-object-api --steps 20 --warmup 3 2> $OUT/bench_${WL}_$n.err | tail -1 >> $OUT/lines_$WL.txt
  done
  python - $OUT/lines_$WL.txt > $F <<PY
import json, sys
runs = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
base = next((r["value"] for r in runs if r["n_gpus"] == 1), None)
print(json.dumps({"metric": runs[0]["metric"] if runs else None, "scaling": "weak", "runs": [{k: r.get(k) for k in ("n_gpus", "value", "ms_per_step", "verified", "data", "rccl_ranks")} for r in runs],
                  "efficiency": {str(r["n_gpus"]): r["value"] / (r["n_gpus"] * base) for r in runs} if base else None}, indent=1))
PY
  cat $F
done

# 4 ---- limb-sharded latency mode, both transports --------------------------------------------------------------------------------
for TR in p2p allgather; do
  n=$(echo $SIZES | awk '{print $NF}')
  python bench.py --gpus $n --workload ckks-limb --limb-transport $TR --no-cpu-baseline --steps 20 --warmup 3 2> $OUT/bench_limb_$TR.err | tail -1 > $OUT/limb_$TR.json
  python -c "import json; r=json.load(open('$OUT/limb_$TR.json')); print('ckks-limb', '$TR', r['n_gpus'], 'ranks', round(r['value']), r['unit'], r['ms_per_step'], 'ms per step', r.get('verified'))"
done

# 5 ---- the C node layer ------------------------------------------------------------------------------------------------------------
if [ -x examples/node_batch ]; then
  # node_batch <ranks> <gpus> [logN] [batch]: rank r runs on device r % gpus
  if [ $TEST_RANKS -gt 1 ]; then examples/node_batch $TEST_RANKS 1 13 16 > $OUT/node_batch.txt 2>&1
  else examples/node_batch $G $G 15 $((32 * G)) > $OUT/node_batch.txt 2>&1; fi
  tail -5 $OUT/node_batch.txt
fi

# 6 ---- RCCL all_gather bandwidth ---------------------------------------------------------------------------------------------------
if [ $G -gt 1 ] && [ $TEST_RANKS -le 1 ]; then
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port 29671 tools/rccl_probe.py > $OUT/rccl_allgather.txt 2>&1
else
  python tools/rccl_probe.py > $OUT/rccl_allgather.txt 2>&1
fi
tail -12 $OUT/rccl_allgather.txt

# 7 ---- hehub's object API over device ranks ------------------------------------------------------------------------------------------
: > $OUT/objapi_scale.txt
for n in $SIZES; do
  if [ $TEST_RANKS -gt 1 ]; then DEVS=$(python -c "print(','.join(['0'] * $n) + (',' if $n == 1 else ''))"); B=$((32 * n)); else DEVS=$n; B=$((256 * n)); fi
  echo "== HEHUB_AMD_DEVICES=$DEVS examples/independent_mults 15 10 $B all 3 8 $((8 * n)) 6" >> $OUT/objapi_scale.txt
  HEHUB_AMD_DEVICES=$DEVS examples/independent_mults 15 10 $B all 3 8 $((8 * n)) 6 >> $OUT/objapi_scale.txt 2>&1
done
grep -E "^==|^serial [0-9]|^batch [0-9]|^chains [0-9]|^devices|digest" $OUT/objapi_scale.txt | grep -v "chain digest" | head -60

# 8 ---- the C node layer's limb-sharded plan under its three transports ---------------------------------------------------------------
if [ $TEST_RANKS -gt 1 ]; then python tools/node_transports.py $TEST_RANKS 4 1 > $OUT/node_transports.txt 2>&1; python tools/node_transports.py 1 4 >> $OUT/node_transports.txt 2>&1
else python tools/node_transports.py $G 8 > $OUT/node_transports.txt 2>&1; fi
cat $OUT/node_transports.txt
ls -la $OUT
