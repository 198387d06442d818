"""The node layer of the C ABI (include/hehub_amd.h "node", hehub_amd/csrc/hp_node.cpp): several ranks behind one handle,
driven from one process.  The test box has one GPU, so the ranks share device 0 (devices = [0, 0, ...]): every code path --
worker threads, per-rank contexts and streams, slicing, staging, the direct peer writes and event ordering of the limb-sharded
mode -- runs as it would on distinct devices, and every result must equal the oracle's words.
Reference: ckks/arith.cpp:55-73, bgv/arith.cpp:59-79, rgsw.cpp:57-156, rescaling.cpp:14-78, mod_switch.cpp:13-78."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu


def make_node(world):
    from hehub_amd.node import Node

    return Node([0] * world)


def case(logn, mext, B, seed):
    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(seed)
    ct1 = rng.poly((B, 2, L, n), mext[:L])
    ct2 = rng.poly((B, 2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    return ct1, ct2, key


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_batch_sharded_host_batches(orc, world):
    """hp_node_ckks_mult_relin_rescale / bgv / ntt on host-resident batches: slices of unequal size, more ranks than items"""
    node = make_node(world)
    try:
        for logn, mext, B in ((12, P.P40[:3] + [P.P50[0]], 7), (7, [P.P40[0], P.P40[1], P.P50[0]], 2)):
            ct1, ct2, key = case(logn, mext, B, 900 + world)
            dk = node.replicate(key)
            cuts = [node.slice(B, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            out = node.ckks_mult(mext, ct1, ct2, dk)
            for i in range(B):
                assert np.array_equal(out[i], orc.ckks_mult(mext, ct1[i], ct2[i], key)), (world, logn, i)
            # the same through page-locked host buffers (hp_host_alloc)
            p1, p2, po = node.pinned(ct1.shape), node.pinned(ct2.shape), node.pinned(out.shape)
            p1[...] = ct1; p2[...] = ct2; po[...] = 0
            node.ckks_mult(mext, p1, p2, dk, out=po)
            assert np.array_equal(po, out)
            for a in (p1, p2, po):
                node.unpin(a)
            out = node.bgv_mult(mext, P.C5_T, ct1, ct2, dk)
            for i in range(B):
                assert np.array_equal(out[i], orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key)), (world, logn, i)
            x = ct1[:, 0].copy()
            node.ntt_(mext[:-1], x)
            for i in range(B):
                assert np.array_equal(x[i], orc.poly_ntt(mext[:-1], ct1[i, 0]))
            node.ntt_(mext[:-1], x, inverse=True, strict=True)
            assert np.array_equal(x, ct1[:, 0])
            node.free_replicas(dk)
    finally:
        node.close()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_batch_sharded_at_parity_level_a(orc, world):
    """hp_node_set_parity_level(A): the batch slices come back as canonical residues = reduce_strict of the oracle's words, and so
    does the limb-sharded mode (round 5: its _strict limb-range stages follow the level -- the digit spread from coefficient WORDS
    that came from other ranks runs on the FP64 kernels; rgsw.cpp:98-153, rescaling.cpp:46-75)"""
    from hehub_amd.node import ShardedPlan

    node = make_node(world)
    try:
        node.set_parity_level("A")
        logn, mext, B = 12, [P.P50[1]] + P.P40[:3] + [P.P50[0]], 5
        L = len(mext) - 1
        ct1, ct2, key = case(logn, mext, B, 1700 + world)
        dk = node.replicate(key)
        q = np.array(mext[:L - 1], dtype=np.uint64)[:, None]
        strict = lambda a: np.where(a >= q, a - q, a)
        out = node.ckks_mult(mext, ct1, ct2, dk)
        for i in range(B):
            assert np.array_equal(out[i], strict(orc.ckks_mult(mext, ct1[i], ct2[i], key))), (world, i)
        out = node.bgv_mult(mext, P.C5_T, ct1, ct2, dk)
        for i in range(B):
            assert np.array_equal(out[i], strict(orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key))), (world, i)
        for t in (0, P.C5_T):
            plan = ShardedPlan(node, logn, mext, 2, plain_modulus=t)
            try:
                res = plan.mult(ct1[:2], ct2[:2], dk)
                for i in range(2):
                    exp = orc.bgv_mult(mext, t, ct1[i], ct2[i], key) if t else orc.ckks_mult(mext, ct1[i], ct2[i], key)
                    assert np.array_equal(res[i], strict(exp)), (world, t, i)
            finally:
                plan.close()
        node.free_replicas(dk)
        node.set_parity_level("B")
        out = node.ckks_mult(mext, ct1, ct2, node.replicate(key))
        assert np.array_equal(out[0], orc.ckks_mult(mext, ct1[0], ct2[0], key))
    finally:
        node.close()


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("logn,mext,B", [(12, [P.P50[1]] + P.P40[:4] + [P.P50[0]], 3), (7, [P.P40[0], P.P40[1], P.P50[0]], 2)])
def test_limb_sharded_matches_the_oracle(orc, world, logn, mext, B):
    """hp_node_sharded_mult: one batch cut by output modulus over `world` ranks (more ranks than moduli included), direct
    peer writes between the ranks' buffers; CKKS and BGV (plain_modulus == 1 quirk of bgv::relinearize included); the
    same plan run twice (buffers are reused) with different inputs"""
    from hehub_amd.node import ShardedPlan

    node = make_node(world)
    try:
        for t in (0, P.C5_T):
            plan = ShardedPlan(node, logn, mext, B, plain_modulus=t)
            owned = [plan.range(r) for r in range(world)]
            assert owned[0][0] == 0 and owned[-1][1] == len(mext) and max(b - a for a, b in owned) - min(b - a for a, b in owned) <= 1
            for rep in range(2):
                ct1, ct2, key = case(logn, mext, B, 70 * world + rep)
                dk = node.replicate(key)
                out = plan.mult(ct1, ct2, dk)
                for i in range(B):
                    exp = orc.bgv_mult(mext, t, ct1[i], ct2[i], key) if t else orc.ckks_mult(mext, ct1[i], ct2[i], key)
                    assert np.array_equal(out[i], exp), (world, t, rep, i)
                node.free_replicas(dk)
            plan.close()
    finally:
        node.close()


def test_limb_sharded_c3_shape(orc):
    """the C3 shape (N = 32768, L = 10: 11 extended moduli over 8 ranks = 2,2,2,1,1,1,1,1) on two ciphertext pairs"""
    from hehub_amd.node import ShardedPlan

    node = make_node(8)
    try:
        mext = P.C3_MODULI_EXT
        plan = ShardedPlan(node, P.C3_LOGN, mext, 2)
        sizes = [plan.range(r)[1] - plan.range(r)[0] for r in range(8)]
        assert sizes == [2, 2, 2, 1, 1, 1, 1, 1] and plan.range(7) == (10, 11)   # the special prime alone on the last rank
        ct1, ct2, key = case(P.C3_LOGN, mext, 2, 33)
        dk = node.replicate(key)
        out = plan.mult(ct1, ct2, dk)
        for i in range(2):
            assert np.array_equal(out[i], orc.ckks_mult(mext, ct1[i], ct2[i], key))
        plan.close()
    finally:
        node.close()


def test_node_errors_are_reported_not_thrown():
    from hehub_amd import capi
    from hehub_amd.engine import HpError
    from hehub_amd.node import Node

    lib = capi.load()
    h = capi.P()
    assert lib.hp_node_create((capi.INT * 1)(99), 1, __import__("ctypes").byref(h)) != capi.HP_OK   # no such device
    node = make_node(2)
    try:
        ct = np.zeros((1, 2, 1, 8), dtype=np.uint64)
        with pytest.raises(HpError):
            node.ckks_mult([P.P40[0], P.P50[0]], ct, ct, node.replicate(np.zeros(8, dtype=np.uint64)))   # L = 1: nothing to drop
    finally:
        node.close()


@pytest.mark.parametrize("world", [2, 3])
def test_device_resident_entry_points(orc, world):
    """hp_node_dev_* (per-rank device slices, unequal counts) and hp_node_sharded_mult_dev (operands replicated on the ranks, the
    whole result left on every rank)"""
    import torch

    from hehub_amd.node import ShardedPlan, dev_mult

    node = make_node(world)
    try:
        logn, mext, B = 12, [P.P50[1]] + P.P40[:3] + [P.P50[0]], 5
        n, L = 1 << logn, len(mext) - 1
        ct1, ct2, key = case(logn, mext, B, 4100 + world)
        dk = node.replicate(key)
        tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
        host = lambda t: t.cpu().numpy().view(np.uint64)
        cuts = [node.slice(B, r) for r in range(world)]
        d1 = [tdev(ct1[lo:hi]) for lo, hi in cuts]; d2 = [tdev(ct2[lo:hi]) for lo, hi in cuts]
        for t in (0, P.C5_T):
            do = [torch.full((hi - lo, 2, L - 1, n), -1, dtype=torch.int64, device="cuda:0") for lo, hi in cuts]
            dev_mult(node, mext, d1, d2, dk, do, plain_modulus=t)
            torch.cuda.synchronize()
            got = np.concatenate([host(x) for x in do])
            for i in range(B):
                exp = orc.bgv_mult(mext, t, ct1[i], ct2[i], key) if t else orc.ckks_mult(mext, ct1[i], ct2[i], key)
                assert np.array_equal(got[i], exp), (world, t, i)
            plan = ShardedPlan(node, logn, mext, B, plain_modulus=t)
            r1 = [tdev(ct1) for _ in range(world)]; r2 = [tdev(ct2) for _ in range(world)]
            ro = [torch.full((B, 2, L - 1, n), -1, dtype=torch.int64, device="cuda:0") for _ in range(world)]
            torch.cuda.synchronize()
            plan.mult_dev(r1, r2, dk, ro)
            for r in range(world):
                assert np.array_equal(host(ro[r]), got), (world, t, r)
            plan.close()
        node.free_replicas(dk)
    finally:
        node.close()


def test_peer_matrix_and_the_staged_exchange(orc, monkeypatch):
    """hp_node_peer_matrix reports which rank pairs write each other directly; with HP_NODE_NO_PEER (the injected "no peer
    access" switch) no pair does and the limb-sharded plan must take its staged branch (owner -> page-locked host -> peer) --
    same words as the direct branch and as the oracle, host- and device-resident entry points."""
    import torch

    from hehub_amd.node import Node, ShardedPlan

    world, logn, mext, B = 3, 12, [P.P50[1]] + P.P40[:3] + [P.P50[0]], 3
    n, L = 1 << logn, len(mext) - 1
    ct1, ct2, key = case(logn, mext, B, 5150)
    results = {}
    for mode in ("direct", "staged"):
        if mode == "staged":
            monkeypatch.setenv("HP_NODE_NO_PEER", "1")
        else:
            monkeypatch.delenv("HP_NODE_NO_PEER", raising=False)
        node = Node([0] * world)
        try:
            m = node.peer_matrix()
            assert m.shape == (world, world) and (np.diag(m) == 1).all()
            assert (m == 1).all() if mode == "direct" else (m == np.eye(world, dtype=np.int32)).all()   # ranks share device 0
            dk = node.replicate(key)
            for t in (0, P.C5_T):
                plan = ShardedPlan(node, logn, mext, B, plain_modulus=t)
                out = plan.mult(ct1, ct2, dk)
                for i in range(B):
                    exp = orc.bgv_mult(mext, t, ct1[i], ct2[i], key) if t else orc.ckks_mult(mext, ct1[i], ct2[i], key)
                    assert np.array_equal(out[i], exp), (mode, t, i)
                tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
                r1 = [tdev(ct1) for _ in range(world)]; r2 = [tdev(ct2) for _ in range(world)]
                ro = [torch.full((B, 2, L - 1, n), -1, dtype=torch.int64, device="cuda:0") for _ in range(world)]
                torch.cuda.synchronize()
                plan.mult_dev(r1, r2, dk, ro)
                for r in range(world):
                    assert np.array_equal(ro[r].cpu().numpy().view(np.uint64), out), (mode, t, r)
                results[(mode, t)] = out
                plan.close()
            node.free_replicas(dk)
        finally:
            node.close()
    for t in (0, P.C5_T):
        assert np.array_equal(results[("direct", t)], results[("staged", t)])


def test_node_leaves_the_callers_device_alone():
    """hp_node_create / hp_ctx_* calls run with their own device current and put the caller's back"""
    import torch

    from hehub_amd.node import Node

    before = torch.cuda.current_device()
    node = Node([0, 0])
    try:
        assert torch.cuda.current_device() == before
        a = node.pinned((4,))
        node.unpin(a)
        assert torch.cuda.current_device() == before
    finally:
        node.close()
    assert torch.cuda.current_device() == before


def test_failing_rank_is_the_one_reported():
    """run_all keeps the message of the rank whose own call failed (taken on that rank's worker thread), not of a rank that
    merely stopped because a peer did: a limb-sharded call with one rank's key pointer NULL"""
    import ctypes as C

    from hehub_amd import capi
    from hehub_amd.node import Node, ShardedPlan

    node = Node([0, 0, 0])
    try:
        logn, mext, B = 11, [P.P40[0], P.P40[1], P.P50[0]], 1
        ct1, ct2, key = case(logn, mext, B, 77)
        plan = ShardedPlan(node, logn, mext, B)
        dk = node.replicate(key)
        keys = (C.c_void_p * 3)(*[dk[r] for r in range(3)])
        keys[2] = None                                   # rank 2 gets no key
        out = np.zeros((B, 2, len(mext) - 2, 1 << logn), dtype=np.uint64)
        rc = node.lib.hp_node_sharded_mult(plan.h, ct1.ctypes.data_as(capi.P), ct2.ctypes.data_as(capi.P), keys, out.ctypes.data_as(capi.P))
        msg = node.lib.hp_node_last_error(node.h).decode()
        assert rc == capi.HP_EINVAL and msg.startswith("rank 2:") and "NULL" in msg, (rc, msg)
        # the node is usable afterwards
        good = plan.mult(ct1, ct2, dk)
        assert good.shape == out.shape
        plan.close()
        node.free_replicas(dk)
    finally:
        node.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_limb_sharded_transports_give_the_same_words(orc, world):
    """the collective the north star names, inside the C node layer.  hp_node_set_transport: direct peer writes
    (default), RCCL (ncclAllGather of the packed coefficient-digit parts / of the result limbs, ncclBroadcast of the dropped limbs'
    coefficients; librccl.so loaded on demand) and the same packed buffers moved by copies.  Every sum over the digits for an output
    modulus is formed on one rank in hehub's order (rgsw.cpp:98-153), so the words are hehub's under every transport.  RCCL needs one
    device per rank: on this one-GPU box it runs with ONE rank (the calls, the packing and the unpacking of a one-rank communicator) and
    must refuse ranks that share the GPU with HP_EUNSUPPORTED; the packed transport covers the multi-rank packing."""
    from hehub_amd import capi
    from hehub_amd.engine import HpError
    from hehub_amd.node import ShardedPlan

    import torch

    node = make_node(world)
    try:
        logn, mext, B = 12, [P.P50[1]] + P.P40[:4] + [P.P50[0]], 3
        n, L = 1 << logn, len(mext) - 1
        assert node.transport() == "peer"
        shared = world > 1 and torch.cuda.device_count() < world
        names = ["peer", "packed"] + ([] if shared else ["rccl"])
        if shared:
            with pytest.raises(HpError) as e:
                node.set_transport("rccl")
            assert e.value.code == capi.HP_EUNSUPPORTED and "own device" in e.value.msg and node.transport() == "peer"
        for t in (0, P.C5_T):
            ct1, ct2, key = case(logn, mext, B, 9100 + world)
            dk = node.replicate(key)
            want = np.stack([orc.bgv_mult(mext, t, ct1[i], ct2[i], key) if t else orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)])
            for name in names:
                node.set_transport(name)
                assert node.transport() == name
                plan = ShardedPlan(node, logn, mext, B, plain_modulus=t)
                for rep in range(2):   # (the plan's buffers are reused)
                    out = plan.mult(ct1, ct2, dk)
                    assert np.array_equal(out, want), (world, t, name, rep)
                # operands replicated on the ranks, the whole result left on EVERY rank (the all-gather of the result limbs)
                tdev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
                r1 = [tdev(ct1) for _ in range(world)]; r2 = [tdev(ct2) for _ in range(world)]
                ro = [torch.full((B, 2, L - 1, n), -1, dtype=torch.int64, device="cuda:0") for _ in range(world)]
                torch.cuda.synchronize()
                plan.mult_dev(r1, r2, dk, ro)
                for r in range(world):
                    assert np.array_equal(ro[r].cpu().numpy().view(np.uint64), want), (world, t, name, r)
                plan.close()
            node.free_replicas(dk)
        node.set_transport("peer")
    finally:
        node.close()
