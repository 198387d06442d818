"""The RCCL side of the multi-GPU plumbing, executed for real on ONE GPU: a torch.distributed process group with backend
"nccl" (= RCCL on ROCm), world size 1, bound to cuda:0, running exactly the calls an N-GPU job makes -- hehub_amd.dist's
barrier / MAX all-reduce on a device tensor, bench.py's MIN all-reduce of the verify flag, and the collectives of the
limb-sharded mode (sharded.Comm: all_gather on device tensors, batched isend / irecv, broadcast).  World size 1 crosses no xGMI
link, but it initialises the communicator (HSA_ENABLE_IPC_MODE_LEGACY handling included) and pushes device buffers through
RCCL's kernels, which no gloo test does."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from hehub_amd import dist as hd
from hehub_amd import sharded

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
t0 = time.perf_counter()
world, rank = hd.init("nccl", device=dev, force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and (world, rank) == (1, 0)
hd.barrier()
init_s = time.perf_counter() - t0
res = {"init_s": init_s, "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
# the timing fences of bench.py
assert hd.max_over_ranks(1.25, device=dev) == 1.25
flag = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
assert int(flag.item()) == 1
# the exchanges of the limb-sharded mode on device tensors (int64 words, the engine's buffers)
g = torch.Generator(device="cpu").manual_seed(7)
buf = torch.randint(-2**62, 2**62, (6, 11, 4096), dtype=torch.int64, generator=g).to(dev)
ref = buf.clone()
os.environ["HP_SHARDED_FORCE_COLLECTIVES"] = "1"
for transport in sharded.Comm.TRANSPORTS:
    comm = sharded.Comm(transport=transport)
    assert comm.world == 1 and comm.rank == 0 and comm.staged is False
    comm.all_gather_limbs(buf, [(0, 11)])
    comm.broadcast(buf, 0)
    torch.cuda.synchronize()
    assert torch.equal(buf, ref), transport
# the raw forms with something to move: all_gather into a list, and a send to / receive from this rank itself in one batch
out = [torch.empty_like(buf)]
dist.all_gather(out, buf)
torch.cuda.synchronize()
assert torch.equal(out[0], ref)
recv = torch.zeros_like(buf)
for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, 0), dist.P2POp(dist.irecv, recv, 0)]):
    req.wait()
torch.cuda.synchronize()
assert torch.equal(recv, ref)
res["collectives"] = ["barrier", "all_reduce MAX f64", "all_reduce MIN i32", "all_gather i64", "batch_isend_irecv self", "broadcast"]
hd.finalize()
print("RCCL_OK " + json.dumps(res))
'''


@pytest.mark.gpu
def test_rccl_one_rank_runs_the_multi_gpu_calls():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    ok = [l for l in out.stdout.splitlines() if l.startswith("RCCL_OK ")]
    assert out.returncode == 0 and len(ok) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    r = json.loads(ok[0][len("RCCL_OK "):])
    assert len(r["collectives"]) == 6


@pytest.mark.gpu
def test_bench_line_reports_an_initialised_rccl_communicator():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ntt", "--logn", "12", "--batch", "64", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    # RCCL's version banner (printed on fd 1 when the communicator is created) must not reach the bench's stdout: nothing but the
    # `#section` lines and, LAST, the contract line
    nonempty = [l for l in out.stdout.splitlines() if l.strip()]
    assert nonempty[-1] == lines[0] and all(l.startswith("#section ") for l in nonempty[:-1]), out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["rccl_ranks"] == 1 and r["rccl"]["initialised"] is True and r["rccl"]["backend"] == "nccl" and r["verified"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["p2p", "allgather"])
def test_limb_sharded_mode_through_rccl_one_rank(transport):
    """bench.py --workload ckks-limb on ONE rank with the exchanges forced through the RCCL communicator (all_gather of the key-switch
    coefficient rows and of the result, broadcast of the dropped limbs' coefficients): the code an N-GPU latency-mode job runs, every
    output verified against the checker"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HP_SHARDED_FORCE_COLLECTIVES"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ckks-limb", "--limb-transport", transport, "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    r = json.loads(lines[0])
    assert r["rccl_ranks"] == 1 and r["verified"] is True and r["config"]["digit_exchange"] == transport and r["scaling"] == "strong"
