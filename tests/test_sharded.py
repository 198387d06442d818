"""Limb-sharded ("latency") mode, hehub_amd/sharded.py (the all-gather of the key-switch digits).

CPU tier: ownership ranges and the two exchanges over a 2-rank gloo group (host tensors).
GPU tier: (a) all virtual ranks of a world of R run in lockstep in ONE process on shared buffers -- this checks
that the limb-range stages of the C ABI compose to exactly the oracle's ciphertext for every cut; (b) two real
processes on the one GPU of the test box exchange through gloo and each compare with the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_limb_ranges_partition():
    from hehub_amd.sharded import clip, limb_ranges, owner_of

    for n in (1, 2, 7, 11, 12):
        for world in (1, 2, 3, 4, 8, 16):
            r = limb_ranges(n, world)
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
            for k in range(n):
                lo, hi = r[owner_of(k, r)]
                assert lo <= k < hi
            c = clip(r, n - 1)
            assert c[-1][1] == n - 1 or n == 1


COMM_WORKER = r"""
import sys
sys.path.insert(0, sys.argv[1])
import torch
from hehub_amd import dist as hd
from hehub_amd.sharded import Comm, limb_ranges, clip
world, rank = hd.init("gloo")
comm = Comm(transport=sys.argv[2] if len(sys.argv) > 2 else None)
assert comm.transport == (sys.argv[2] if len(sys.argv) > 2 else "p2p")
ranges = clip(limb_ranges(4, world), 3)          # 3 limbs over 2 ranks: (0,2) and (2,3)
full = torch.arange(5 * 3 * 8, dtype=torch.int64).reshape(5, 3, 8) * 7 + 1
buf = torch.zeros_like(full)
lo, hi = ranges[rank]
buf[:, lo:hi] = full[:, lo:hi]
comm.all_gather_limbs(buf, ranges)
assert torch.equal(buf, full), (rank, buf)
t = full[0].clone() if rank == 1 else torch.zeros_like(full[0])
comm.broadcast(t, 1)
assert torch.equal(t, full[0])
hd.barrier(sync_device=False)
hd.finalize()
print("rank", rank, "ok")
"""


def _spawn(script_text, tmp_path, extra_args=(), timeout=600):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, *extra_args], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1], outs


@pytest.mark.parametrize("transport", ["p2p", "allgather"])
def test_exchanges_two_rank_gloo(tmp_path, transport):
    """both transports of the digit exchange: batched isend / irecv of exact slices, and ONE all_gather of padded slices (the
    collective BASELINE's north star names)"""
    _spawn(COMM_WORKER, tmp_path, extra_args=(transport,))


def _case(orc, logn, mext, B, seed):
    from oracle.pyoracle import SplitMix

    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(seed)
    ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    key = rng.poly((L, 2, L + 1, n), mext)
    return ct1, ct2, key


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["B", "A"])
@pytest.mark.parametrize("logn,nmod", [(11, 4), (12, 3), (5, 4), (13, 6)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_virtual_ranks_compose_to_the_oracle(orc, logn, nmod, world, level):
    """level A (round 5): the limb-range stages follow the context's parity level -- every output word is reduce_strict of the
    oracle's word (a ring degree without tiled kernels runs at level B and returns the raw words)"""
    from hehub_amd.engine import Engine
    from hehub_amd.sharded import ShardedMult

    mext = P.P40[:nmod - 1] + [P.P50[0]]       # nmod - 1 ciphertext moduli + the special prime
    B = 2
    ct1, ct2, key = _case(orc, logn, mext, B, 100 + logn)
    eng = Engine(0)
    eng.set_parity_level(level)
    qs = np.array(mext[:nmod - 2], dtype=np.uint64)[:, None]
    fin = (lambda a: np.where(a >= qs, a - qs, a)) if (level == "A" and 11 <= logn <= 15) else (lambda a: a)
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    for t in (0, 65537):
        sm = ShardedMult(eng, mext, world, plain_modulus=t)
        bufs = sm.buffers(B, 1 << logn)
        for b in bufs.values():
            b.fill_(-1)                      # garbage: every word of the result must be produced by some rank
        gens = [sm.stages(r, d1, d2, dk, bufs) for r in range(world)]
        alive = True
        while alive:                          # lockstep: all ranks reach the same exchange; buffers are shared, so it is a no-op
            reqs = [next(g, None) for g in gens]
            alive = any(r is not None for r in reqs)
            assert all((r is None) == (reqs[0] is None) for r in reqs)
        got = eng.to_host(bufs["out"])
        exp = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) if t == 0 else orc.bgv_mult(mext, t, ct1[i], ct2[i], key)
                        for i in range(B)])
        assert np.array_equal(got, fin(exp)), (logn, nmod, world, t, level)
    eng.close()


GPU_WORKER = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np, torch
import params as P
from hehub_amd import dist as hd
from hehub_amd.engine import Engine
from hehub_amd.sharded import Comm, ShardedMult
from oracle.pyoracle import Oracle, SplitMix
world, rank = hd.init("gloo")
torch.cuda.set_device(0)
orc, eng, comm = Oracle("orc"), Engine(0), Comm(transport=sys.argv[2] if len(sys.argv) > 2 else None)
logn, mext, B = 12, P.P40[:4] + [P.P50[0]], 2
n, L = 1 << logn, len(mext) - 1
rng = SplitMix(77)                         # same seed on both ranks: the ciphertexts are replicated
ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)]); ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
key = rng.poly((L, 2, L + 1, n), mext)
d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
for t in (0, 65537):
    sm = ShardedMult(eng, mext, world, plain_modulus=t)
    bufs = sm.buffers(B, n)
    for b in bufs.values(): b.fill_(-1)
    out = sm.run(comm, d1, d2, dk, bufs)
    exp = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) if t == 0 else orc.bgv_mult(mext, t, ct1[i], ct2[i], key) for i in range(B)])
    assert np.array_equal(eng.to_host(out), exp), (rank, t)
hd.barrier()
hd.finalize()
eng.close()
print("rank", rank, "ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["p2p", "allgather"])
def test_two_processes_one_gpu_gloo(tmp_path, transport):
    _spawn(GPU_WORKER, tmp_path, extra_args=(transport,), timeout=900)
