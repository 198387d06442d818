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
