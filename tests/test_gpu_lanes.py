"""Context families of the C ABI (round 5; include/hehub_amd.h: hp_ctx_fork, hp_ctx_wait_for, hp_dev_gather_rows / scatter_rows): lanes
of one GPU with their own stream and scratch workspace and ONE copy of the tables.  hehub's callers loop over independent single-ciphertext
calls (src/circuits/linear_algebra.h:109-133); the host layer spreads those over lanes -- here the primitives themselves, through ctypes,
against the oracle: concurrent calls on three lanes, a cross-lane dependency ordered by hp_ctx_wait_for, destruction in any order."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu


def test_forked_lanes_run_concurrently_and_in_order(orc):
    import torch

    from hehub_amd.engine import Engine

    root = Engine(0, use_torch_stream=False)     # own streams: the lanes must not serialise on torch's current stream
    lanes = [root, root.fork(), root.fork()]
    try:
        logn, mext = 13, [P.P50[1]] + P.P40[:3] + [P.P50[0]]
        n, L = 1 << logn, len(mext) - 1
        rng = SplitMix(5150)
        key = rng.poly((L, 2, L + 1, n), mext)
        dk = root.to_device(key)
        cts = [(rng.poly((2, 2, L, n), mext[:L]), rng.poly((2, 2, L, n), mext[:L])) for _ in lanes]
        dev = [(root.to_device(a), root.to_device(b)) for a, b in cts]
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):                       # interleaved: every lane has work in flight at the same time
            outs = [e.ckks_mult(mext, a, b, dk) for e, (a, b) in zip(lanes, dev)]
        # a dependency across lanes: lane 1 adds what lane 2 produced -- only correct behind hp_ctx_wait_for
        lanes[1].wait_for(lanes[2])
        s = lanes[1].poly_add(mext[:L - 1], outs[2].reshape(4, L - 1, n), outs[2].reshape(4, L - 1, n))
        for e in lanes:
            e.sync()
        for (a, b), o in zip(cts, outs):
            got = root.to_host(o)
            for i in range(2):
                assert np.array_equal(got[i], orc.ckks_mult(mext, a[i], b[i], key))
        exp = np.stack([orc.poly_add(mext[:L - 1], x, x) for x in root.to_host(outs[2]).reshape(4, L - 1, n)])
        assert np.array_equal(root.to_host(s), exp)
        # gather / scatter of rows that lie anywhere: the packing step of the batched forms of hehub_amd_ext.hpp
        rows = [outs[i % 3][i % 2, (i // 2) % 2].contiguous() for i in range(70)]      # more than 64 rows: two launches
        packed = root.empty((70, L - 1, n))
        root.gather_rows(rows, (L - 1) * n, packed)
        root.sync()
        assert all(torch.equal(packed[i], rows[i]) for i in range(70))
        back = [torch.zeros_like(r) for r in rows]
        root.scatter_rows(packed, (L - 1) * n, back)
        root.sync()
        assert all(torch.equal(back[i], rows[i]) for i in range(70))
    finally:
        lanes[1].close()          # a child first, then the parent, then the last child: the family's tables go with the last member
        root.close()
        lanes[2].close()


def test_fork_shares_tables_and_keeps_own_workspace():
    from hehub_amd.engine import Engine

    root = Engine(0)
    child = root.fork()
    try:
        q = P.P40[:2]
        x = np.arange(2 * 4096, dtype=np.uint64).reshape(1, 2, 4096) % np.uint64(q[0])
        a = root.to_host(root.ntt_(q, root.to_device(x)))
        y = child.ntt_(q, child.to_device(x))                      # tables were built by the parent's call
        child.sync()                                                # (the lane has its own stream: torch's copy must not overtake it)
        b = child.to_host(y)
        assert np.array_equal(a, b)
        assert child.parity_level() == root.parity_level() == "B"
        child.set_parity_level("A")                                # per lane
        assert child.parity_level() == "A" and root.parity_level() == "B"
        assert root.lib.hp_ctx_wait_for(child.h, None) != 0 and root.lib.hp_ctx_fork(None, None) != 0   # argument errors, not crashes
    finally:
        child.close()
        root.close()


def strict(moduli, a):
    """reduce_strict per limb (mod_arith.h:58-72): a [..., L, n] of lazy words below 2q"""
    q = np.array(moduli, dtype=np.uint64)[:, None]
    return np.where(a >= q, a - q, a)


def test_family_members_driven_by_their_own_threads(orc):
    """four members of one family, each called from its own host thread AT THE SAME TIME, each touching moduli / ring degrees the
    family has not seen yet (tables, plans, FP64 tables of level A, permutations are built under the family's one mutex while the
    other threads are inside their own calls): every result is the oracle's"""
    import threading

    import torch

    from hehub_amd.engine import Engine

    root = Engine(0, use_torch_stream=False)
    members = [root, root.fork(), root.fork(), root.fork()]
    members[3].set_parity_level("A")
    rng = SplitMix(777)
    work = []
    for t, logn in enumerate((12, 13, 11, 12)):
        mext = [P.P50[1]] + P.P40[t:t + 3] + [P.P50[0]]           # a different set of primes per thread
        n, L = 1 << logn, len(mext) - 1
        work.append((logn, mext, rng.poly((L, 2, L + 1, n), mext), rng.poly((3, 2, L, n), mext[:L]), rng.poly((3, 2, L, n), mext[:L])))
    got, errors = [None] * 4, []
    start = threading.Barrier(4)

    def body(t):
        try:
            e, (logn, mext, key, a, b) = members[t], work[t]
            L = len(mext) - 1
            start.wait()
            dk, da, db = e.to_device(key), e.to_device(a), e.to_device(b)
            # These engines own their streams (use_torch_stream=False) while the tensors come from torch's caching allocator, which hands
            # a freed block to the next allocation at once -- of ANY thread: every tensor an engine call touches stays referenced until
            # that engine has been synchronised, and the copy torch makes on its own stream is waited for before the engine reads it.
            keep = []
            for _ in range(4):
                out = e.ckks_mult(mext, da, db, dk)
                rot = e.ckks_rotate(mext, da, dk, 1)
                src = da.clone().reshape(6, L, 1 << logn)
                torch.cuda.current_stream().synchronize()
                x = e.ntt_(mext[:L], e.intt_(mext[:L], src))
                keep.append((out, rot, src, x))
            e.sync()
            got[t] = (e.to_host(out), e.to_host(rot), e.to_host(x))
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=body, args=(t,)) for t in range(4)]
    try:
        for th in threads:
            th.start()
        for th in threads:
            th.join(600)
        assert not errors, errors
        for t, (logn, mext, key, a, b) in enumerate(work):
            L, n = len(mext) - 1, 1 << logn
            out, rot, x = got[t]
            for i in range(3):
                want = orc.ckks_mult(mext, a[i], b[i], key)
                want_rot = orc.ckks_rotate(mext, a[i], key, 1)
                if t == 3:      # level A: the canonical residue of hehub's word
                    want, want_rot = strict(mext[:L - 1], want), strict(mext[:L], want_rot)
                assert np.array_equal(out[i], want), (t, i)
                assert np.array_equal(rot[i], want_rot), (t, i)
            rt = np.stack([orc.poly_ntt(mext[:L], orc.poly_intt(mext[:L], p)) for p in a.reshape(6, L, n)])
            assert np.array_equal(x, rt), t
    finally:
        for e in members[1:]:
            e.close()
        root.close()
