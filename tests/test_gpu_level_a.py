"""Parity level A (opt-in: hp_ctx_set_parity_level / HP_PARITY_LEVEL=A, include/hehub_amd.h): the scheme-level pipelines on the
FP64 residue transforms of hp_ntt_a.hip.  Contract ("Parity levels"): every output word is congruent to
the reference's word modulo its q -- here: EQUAL to reduce_strict (mod_arith.h:58-72) of the oracle's raw word, since every
entry point that ends in a drop-last-prime returns canonical residues; the key switch alone returns lazy words (< 2q) with the
reference's residues.  Reference contracts: ntt.cpp:155-175, :178-223, rgsw.cpp:98-153, rescaling.cpp:46-75,
mod_switch.cpp:45-77.  The NTT / mod-arith primitives must stay bit-exact (level B) whatever the context's level."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix
from test_gpu_full_batch import PERIOD, all_equal_classes, tile
from test_gpu_parity import _ntt_primes

pytestmark = pytest.mark.gpu
U = np.uint64


@pytest.fixture(scope="module")
def enga():
    from hehub_amd.engine import Engine

    e = Engine(0)
    e.set_parity_level("A")
    assert e.parity_level() == "A"
    yield e
    e.release_workspace()
    e.close()


def strict(moduli, a):
    """reduce_strict per limb: a [..., L, n] of lazy words below 2q"""
    q = np.array(moduli, dtype=U)[:, None]
    return np.where(a >= q, a - q, a)


def canon(moduli, a):
    """the canonical residue of any u64 word of a [..., L', n] array whose limbs are the first L' of `moduli`"""
    return a % np.array(moduli[:a.shape[-2]], dtype=U)[:, None]


CASES = [
    (11, P.P40[:3] + [P.P50[0]], 2),                      # narrow limbs + a wide special prime
    (12, [P.P50[1]] + P.P40[:4] + [P.P50[0]], 3),         # the C3 chain's shape: wide q0, narrow q1.., wide p
    (P.C5_LOGN, P.C5_MODULI_EXT, 2),                      # C5: all narrow
    (14, P.P50[1:4] + [P.P50[0]], 2),                     # all wide (50-bit)
    (15, [P.P50[1]] + P.P40[:2] + [P.P50[0]], 1),         # N = 32768 (lane-pair swap + padded exchange)
    (13, _ntt_primes(2, 13, 44) + _ntt_primes(2, 13, 45) + _ntt_primes(1, 13, 49), 2),   # either side of the wide threshold 2^44
    (3, [1099510054913, 1073479681, 1072496641, 1099507695617], 2),   # no tiled kernels at this degree: runs at level B
]


@pytest.mark.parametrize("logn,mext,B", CASES)
def test_scheme_level_residues(enga, orc, logn, mext, B):
    eng = enga
    n, L = 1 << logn, len(mext) - 1
    q = mext[:L]
    tiled = 11 <= logn <= 15
    rng = SplitMix(7000 + logn)
    ct1 = rng.poly((B, 2, L, n), q)
    ct2 = rng.poly((B, 2, L, n), q)
    key = rng.poly((L, 2, L + 1, n), mext)
    # worst-case magnitudes: the largest canonical word and the largest lazy word in some coefficients
    ct1[0, 0, :, :7] = (np.array(q, dtype=U) - U(1))[:, None]
    ct2[0, 1, :, :5] = (U(2) * np.array(q, dtype=U) - U(1))[:, None]
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    exp = lambda f: np.stack([f(i) for i in range(B)])
    fin = (lambda m, a: strict(m, a)) if tiled else (lambda m, a: a)          # what a drop returns at level A / B
    quad = eng.mult_low_level(q, d1, d2)
    quad_h = eng.to_host(quad)
    assert np.array_equal(quad_h, exp(lambda i: orc.mult_low_level(q, ct1[i], ct2[i])))    # not a transform: level B words
    ext = eng.ext_prod(mext, quad[:, 2].contiguous(), dk)
    ext_h = eng.to_host(ext)
    ext_o = exp(lambda i: orc.ext_prod(mext, quad_h[i, 2], key))
    assert np.array_equal(canon(mext, ext_h), canon(mext, ext_o))
    assert (ext_h < U(2) * np.array(mext, dtype=U)[:, None]).all()
    if not tiled:
        assert np.array_equal(ext_h, ext_o)
    # drops of engine-made and of caller-made ciphertexts
    assert np.array_equal(eng.to_host(eng.ckks_rescale(mext, ext)), fin(q, exp(lambda i: orc.ckks_rescale(mext, ext_o[i]))))
    assert np.array_equal(eng.to_host(eng.ckks_rescale(q, d1)), fin(q[:-1], exp(lambda i: orc.ckks_rescale(q, ct1[i]))))
    for t in (65537, 2, 1):
        assert np.array_equal(eng.to_host(eng.bgv_mod_switch(q, t, d2)), fin(q[:-1], exp(lambda i: orc.bgv_mod_drop(q, t, ct2[i]))))
    assert np.array_equal(eng.to_host(eng.ckks_relinearize(mext, quad, dk)), fin(q, exp(lambda i: orc.ckks_relinearize(mext, quad_h[i], key))))
    assert np.array_equal(eng.to_host(eng.bgv_relinearize(mext, quad, dk)), fin(q, exp(lambda i: orc.bgv_relinearize(mext, quad_h[i], key))))
    for step in (1, 3):
        assert np.array_equal(eng.to_host(eng.ckks_rotate(mext, d1, dk, step)), fin(q, exp(lambda i: orc.ckks_rotate(mext, ct1[i], key, step))))
    assert np.array_equal(eng.to_host(eng.ckks_conjugate(mext, d2, dk)), fin(q, exp(lambda i: orc.ckks_conjugate(mext, ct2[i], key))))
    assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), fin(q[:-1], exp(lambda i: orc.ckks_mult(mext, ct1[i], ct2[i], key))))
    assert np.array_equal(eng.to_host(eng.bgv_mult(mext, P.C5_T, d1, d2, dk)),
                          fin(q[:-1], exp(lambda i: orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key))))


def test_primitives_stay_level_b(enga, orc):
    """hp_dev_ntt / hp_dev_intt / the host primitives are bit-exact with ntt.cpp:145-223 whatever the context's level"""
    eng = enga
    for logn, moduli in ((12, P.P40[:2]), (15, [P.C3_P, P.P40[0]])):
        rng = SplitMix(77 + logn)
        a = rng.poly((2, len(moduli), 1 << logn), moduli)
        y = eng.to_host(eng.ntt_(moduli, eng.to_device(a)))
        assert np.array_equal(y, np.stack([orc.poly_ntt(moduli, a[i]) for i in range(2)]))
        z = eng.to_host(eng.intt_(moduli, eng.to_device(y)))
        assert np.array_equal(z, np.stack([orc.poly_intt(moduli, y[i]) for i in range(2)]))
    x = SplitMix(5).words(1 << 15, P.C3_P)
    assert np.array_equal(eng.host_ntt(15, P.C3_P, x), orc.ntt(15, P.C3_P, x))


def test_chain_with_a_59_bit_modulus_runs_at_level_b(enga, orc):
    """a modulus >= 2^50 has no FP64 kernels: the call silently keeps level B (raw words)"""
    eng = enga
    logn, mext, B = 11, _ntt_primes(2, 11, 40) + _ntt_primes(1, 11, 59), 1
    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(4)
    ct1, ct2, key = rng.poly((B, 2, L, n), mext[:L]), rng.poly((B, 2, L, n), mext[:L]), rng.poly((L, 2, L + 1, n), mext)
    out = eng.to_host(eng.ckks_mult(mext, eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)))
    assert np.array_equal(out[0], orc.ckks_mult(mext, ct1[0], ct2[0], key))


def test_c3_batch_256_residues(enga, orc):
    """BASELINE config 3 (= C4 per GPU) at level A: all 256 outputs are the strict residues of the oracle's outputs; rotation too"""
    eng = enga
    logn, mext, B = P.C3_LOGN, P.C3_MODULI_EXT, P.C3_BATCH
    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(3)
    b1 = rng.poly((PERIOD, 2, L, n), mext[:L])
    b2 = rng.poly((PERIOD, 2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    ct1, ct2, dk = tile(eng, b1, B), tile(eng, b2, B), eng.to_device(key)
    out = eng.ckks_mult(mext, ct1, ct2, dk)
    exp = np.stack([strict(mext[:L - 1], orc.ckks_mult(mext, b1[c], b2[c], key)) for c in range(PERIOD)])
    assert all_equal_classes(eng, out, exp)
    out.fill_(-1)
    eng.ckks_mult(mext, ct1, ct2, dk, out=out)
    assert all_equal_classes(eng, out, exp)
    rot = eng.ckks_rotate(mext, ct1, dk, 5)
    assert all_equal_classes(eng, rot, np.stack([strict(mext[:L], orc.ckks_rotate(mext, b1[c], key, 5)) for c in range(PERIOD)]))


def test_c5_batch_512_residues(enga, orc):
    """BASELINE config 5 per GPU at level A: bgv mult_low_level + relinearize (plain_modulus == 1 quirk) + mod_switch, 512 pairs"""
    eng = enga
    logn, mext, t, B = P.C5_LOGN, P.C5_MODULI_EXT, P.C5_T, 512
    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(5)
    b1 = rng.poly((PERIOD, 2, L, n), mext[:L])
    b2 = rng.poly((PERIOD, 2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    ct1, ct2, dk = tile(eng, b1, B), tile(eng, b2, B), eng.to_device(key)
    out = eng.bgv_mult(mext, t, ct1, ct2, dk)
    exp = np.stack([strict(mext[:L - 1], orc.bgv_mult(mext, t, b1[c], b2[c], key)) for c in range(PERIOD)])
    assert all_equal_classes(eng, out, exp)


def test_level_a_equals_strict_of_level_b_on_distinct_inputs(enga):
    """all-distinct random C3 ciphertexts (no periodic classes): level A == reduce_strict(level B) word for word on the device"""
    import torch

    from hehub_amd.engine import Engine
    from test_gpu_full_batch import rand_dev

    logn, mext, B = P.C3_LOGN, P.C3_MODULI_EXT, 64
    n, L = 1 << logn, len(mext) - 1
    eb = Engine(0)
    try:
        ct1 = rand_dev(eb, (B, 2, L, n), mext[:L], 131)
        ct2 = rand_dev(eb, (B, 2, L, n), mext[:L], 132)
        key = rand_dev(eb, (L, 2, L + 1, n), mext, 133)
        ob = eb.ckks_mult(mext, ct1, ct2, key)
        eb.poly_reduce_strict_(mext[:L - 1], ob.view(B * 2, L - 1, n))
        oa = enga.ckks_mult(mext, ct1, ct2, key)
        assert torch.equal(oa, ob)
    finally:
        eb.release_workspace()
        eb.close()


@pytest.mark.parametrize("logn", [11, 12, 13, 14, 15])
def test_residue_transforms(enga, orc, logn):
    """hp_dev_ntt_residues / hp_dev_intt_residues: the FP64 transforms as explicit entry points.  Forward == the oracle's lazy words
    modulo q (ntt.cpp:145-176); inverse == the words of intt_negacyclic_inplace (ntt.h:88-92), bit for bit; round trip == input."""
    from hehub_amd.engine import Engine, HpError

    moduli = [P.P50[1], P.P40[0], P.P40[3], P.P50[0]] + (_ntt_primes(1, logn, 44) + _ntt_primes(1, logn, 45))
    n, B = 1 << logn, 3
    rng = SplitMix(900 + logn)
    a = rng.poly((B, len(moduli), n), moduli)
    a[0, :, :9] = (np.array(moduli, dtype=U) - U(1))[:, None]                 # largest canonical words
    lazy = a.copy()
    lazy[1, :, :9] = (U(2) * np.array(moduli, dtype=U) - U(1))[:, None]       # largest lazy words hehub hands over
    for eng in (enga, Engine(0)):                                             # whatever the context's level
        y = eng.to_host(eng.ntt_residues_(moduli, eng.to_device(lazy)))
        exp = np.stack([orc.poly_ntt(moduli, lazy[i]) for i in range(B)])
        assert np.array_equal(y, canon(moduli, exp))
        yl = np.stack([orc.poly_ntt(moduli, a[i]) for i in range(B)])          # the reference's lazy forward words as inverse input
        z = eng.to_host(eng.intt_residues_(moduli, eng.to_device(yl)))
        assert np.array_equal(z, np.stack([orc.poly_reduce_strict(moduli, orc.poly_intt(moduli, yl[i])) for i in range(B)]))
        assert np.array_equal(z, a)
        rt = eng.to_host(eng.intt_residues_(moduli, eng.ntt_residues_(moduli, eng.to_device(a))))
        assert np.array_equal(rt, a)
        if eng is not enga:
            with pytest.raises(HpError, match="below 2\\^50"):
                eng.ntt_residues_([P.C1_Q], eng.empty((1, 1, 1 << 12)))
            with pytest.raises(HpError, match="2\\^11"):
                eng.ntt_residues_([P.P40[0]], eng.empty((1, 1, 1 << 8)))
            eng.close()


@pytest.mark.parametrize("logn,L0,L", [(11, 5, 3), (13, 6, 5)])
def test_extensions_at_level_a(enga, orc, logn, L0, L):
    """keys of a higher level (hp_dev_*_at) and rescale by several primes (hp_dev_ckks_rescale_n) at parity level A: the residues of what
    the oracle gives for the extracted sub-key / for successive single drops (tests/test_extensions.py pins the words at level B)"""
    eng = enga
    n, B = 1 << logn, 2
    q_full, p = P.P40[:L0], P.P50[0]
    rng = SplitMix(1900 + logn)
    key_full = rng.poly((L0, 2, L0 + 1, n), q_full + [p])
    sub = np.ascontiguousarray(key_full[:L][:, :, list(range(L)) + [L0], :])
    mext = q_full[:L] + [p]
    ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key_full)
    exp = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], sub) for i in range(B)])
    assert np.array_equal(eng.to_host(eng.ckks_mult_at(mext, L0, d1, d2, dk)), canon(mext, exp))
    exp = np.stack([orc.ckks_rotate(mext, ct1[i], sub, 3) for i in range(B)])
    assert np.array_equal(eng.to_host(eng.ckks_rotate_at(mext, L0, d1, dk, 3)), canon(mext, exp))
    drops = 2
    q = mext[:L]
    exp = ct1
    for d in range(drops):
        exp = np.stack([orc.ckks_rescale(q[:L - d], exp[i]) for i in range(B)])
    # (the second drop consumes the first one's canonical words: the same residues as the oracle's chain of lazy words)
    assert np.array_equal(eng.to_host(eng.ckks_rescale_n(q, d1, drops)), canon(q, exp))


@pytest.mark.parametrize("logn,mext", [
    (11, [P.P50[1], P.P40[0], P.P50[0]]),          # L = 2: one output limb (wide), narrow q', wide p
    (12, P.P40[:2] + [P.P40[6]]),                   # L = 2, all narrow
    (13, [P.P40[0], P.P50[1], P.P50[2]]),           # L = 2, wide q' and p
    (15, [P.P50[1], P.P40[0], P.P40[1], P.P50[0]]),  # N = 32768 (no spills there), L = 3
    (14, P.P50[1:4] + P.P40[:2] + [P.P50[0]]),      # L = 5, wide limbs first
])
def test_two_drops_in_one_transform(orc, monkeypatch, logn, mext):
    """hp_dev_*_mult_relin_* at level A drops the special prime and the last prime in one transform per output limb
    (hp_api_scheme.cpp: drop_two_last_a): the oracle's residues, the same words as with HP_NO_DOUBLE_DROP (two launches), for CKKS,
    BGV (rescaling.cpp:46-75, mod_switch.cpp:45-77) and the inner_t extension"""
    from hehub_amd.engine import Engine

    n, L, B, t = 1 << logn, len(mext) - 1, 3, P.C5_T
    q = mext[:L]
    rng = SplitMix(4400 + logn)
    ct1, ct2 = rng.poly((B, 2, L, n), q), rng.poly((B, 2, L, n), q)
    key = rng.poly((L, 2, L + 1, n), mext)
    ct1[0, 0, :, :7] = (np.array(q, dtype=U) - U(1))[:, None]
    ct2[0, 1, :, :5] = (U(2) * np.array(q, dtype=U) - U(1))[:, None]
    exp = {"ckks": np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)]),
           "bgv": np.stack([orc.bgv_mult(mext, t, ct1[i], ct2[i], key) for i in range(B)]),
           "bgv_t": np.stack([orc.bgv_mod_drop(q, t, orc.bgv_relinearize(mext, orc.mult_low_level(q, ct1[i], ct2[i]), key, inner_t=t))
                              for i in range(B)])}
    got = {}
    for off in (False, True):
        monkeypatch.delenv("HP_NO_DOUBLE_DROP", raising=False)
        if off:
            monkeypatch.setenv("HP_NO_DOUBLE_DROP", "1")
        eng = Engine(0)
        try:
            eng.set_parity_level("A")
            d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
            got[off] = {"ckks": eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), "bgv": eng.to_host(eng.bgv_mult(mext, t, d1, d2, dk)),
                        "bgv_t": eng.to_host(eng.bgv_mult(mext, t, d1, d2, dk, inner_t=True))}
        finally:
            eng.close()
    for name in exp:
        assert np.array_equal(got[False][name], canon(mext, exp[name])), name
        assert np.array_equal(got[True][name], got[False][name]), name


@pytest.mark.parametrize("logn", [11, 13, 15])
def test_forty_bit_digit_rows(orc, monkeypatch, logn):
    """HP_PACK40 (hp_device.h): at level A the digit rows of a modulus with q + 2 <= 2^40 cross HBM as 5 bytes per word in an offset
    representation.  The largest NTT prime below 2^40 (the offset words reach q + 1), a 30-bit one, one just above 2^40 (stays on 48-bit
    rows) and a 50-bit one in one chain: the oracle's residues for the key switch, rotation and both mult pipelines
    (rgsw.cpp:98-153), and the same words as with HP_NO_PACK40"""
    from hehub_amd.engine import Engine

    top40, bit30, over40 = P.ntt_primes(1, logn, 40)[0], P.ntt_primes(1, logn, 30)[0], P.ntt_primes(1, logn, 41)[0]
    assert top40 + 2 <= 1 << 40 < over40
    mext = [top40, bit30, over40, P.P40[1], P.P50[0]]
    n, L, B = 1 << logn, len(mext) - 1, 2 if logn == 15 else 3
    q = mext[:L]
    rng = SplitMix(5100 + logn)
    ct1, ct2 = rng.poly((B, 2, L, n), q), rng.poly((B, 2, L, n), q)
    key = rng.poly((L, 2, L + 1, n), mext)
    ct1[0, 1, :, :9] = (np.array(q, dtype=U) - U(1))[:, None]
    exp = {"ext": np.stack([orc.ext_prod(mext, ct1[i, 1], key) for i in range(B)]),
           "rot": np.stack([orc.ckks_rotate(mext, ct1[i], key, 5) for i in range(B)]),
           "ckks": np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)]),
           "bgv": np.stack([orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key) for i in range(B)])}
    got = {}
    for off in (False, True):
        monkeypatch.delenv("HP_NO_PACK40", raising=False)
        if off:
            monkeypatch.setenv("HP_NO_PACK40", "1")
        eng = Engine(0)
        try:
            eng.set_parity_level("A")
            d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
            got[off] = {"ext": eng.to_host(eng.ext_prod(mext, d1[:, 1].contiguous(), dk)), "rot": eng.to_host(eng.ckks_rotate(mext, d1, dk, 5)),
                        "ckks": eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), "bgv": eng.to_host(eng.bgv_mult(mext, P.C5_T, d1, d2, dk))}
        finally:
            eng.close()
    for name in exp:
        assert np.array_equal(canon(mext, got[False][name]), canon(mext, exp[name])), name
        if name != "ext":   # everything that ends in a drop is canonical; the key switch alone returns lazy words
            assert np.array_equal(got[False][name], canon(mext, exp[name])), name
            assert np.array_equal(got[True][name], got[False][name]), name
        else:
            assert np.array_equal(canon(mext, got[True][name]), canon(mext, got[False][name])), name


def test_level_a_pipeline_in_a_hip_graph(enga, orc):
    """after one warm-up call (which builds the FP64 tables) a level-A entry point only enqueues kernels: capturable and replayable"""
    import torch

    eng = enga
    logn, L = 12, 3
    mext = [P.P50[1]] + P.P40[:L - 1] + [P.P50[0]]
    n, B = 1 << logn, 2
    rng = SplitMix(2777)
    ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)]); ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    key = rng.poly((L, 2, L + 1, n), mext)
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    out = eng.empty((B, 2, L - 1, n))
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    try:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.use_stream(side)
            eng.ckks_mult(mext, d1, d2, dk, out=out)
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                eng.ckks_mult(mext, d1, d2, dk, out=out)
        torch.cuda.current_stream().wait_stream(side)
        exp = lambda a, b: canon(mext, np.stack([orc.ckks_mult(mext, a[i], b[i], key) for i in range(B)]))
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(eng.to_host(out), exp(ct1, ct2))
        d1.copy_(eng.to_device(ct2)); d2.copy_(eng.to_device(ct1))
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(eng.to_host(out), exp(ct2, ct1))
    finally:
        eng.use_stream(torch.cuda.current_stream())


def test_debug_switches_keep_a_call_at_level_b(orc):
    """hp_ctx_set_force_generic (the simple kernels as cross-check) with the context at level A: the call runs at level B as a whole --
    never a mix of canonical and lazy representatives -- and goes back to level A when the switch is released"""
    from hehub_amd.engine import Engine

    eng = Engine(0)
    try:
        eng.set_parity_level("A")
        logn, mext, B = 11, P.P40[:3] + [P.P50[0]], 2
        n, L = 1 << logn, 3
        rng = SplitMix(31337)
        ct1, ct2 = rng.poly((B, 2, L, n), mext[:L]), rng.poly((B, 2, L, n), mext[:L])
        key = rng.poly((L, 2, L + 1, n), mext)
        d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
        raw = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)])
        assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), canon(mext, raw))
        eng.force_generic(True)
        assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), raw)
        eng.force_generic(False)
        assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), canon(mext, raw))
    finally:
        eng.close()


def test_all_max_lazy_words_on_wide_moduli(enga, orc):
    """four 50-bit moduli (the C2 shape), EVERY input word the largest lazy word 2q - 1 (and a second batch member
    alternating 2q - 1 / 0, the input that grows fastest through the butterflies): without the reduction on load the first forward
    pass leaves the exact FP64 range (hp_ntt_a.hip: B' = B (1 + q 2^-52) + q/2 from 2^51).  Residues of ntt.cpp:155-175, :178-223."""
    eng = enga
    for logn in (11, 14, 15):
        moduli = P.P50[:4]
        n, L = 1 << logn, 4
        top = (U(2) * np.array(moduli, dtype=U) - U(1))[:, None]
        a = np.empty((3, L, n), dtype=U)
        a[0] = top
        a[1] = np.where(np.arange(n)[None, :] % 2 == 0, top, U(0))
        a[2] = SplitMix(880 + logn).poly((L, n), moduli) + np.array(moduli, dtype=U)[:, None] * (np.arange(n)[None, :] % 2).astype(U)   # lazy: some words in [q, 2q)
        fwd = np.stack([orc.poly_ntt(moduli, a[i]) for i in range(3)])
        y = eng.to_host(eng.ntt_residues_(moduli, eng.to_device(a)))
        assert np.array_equal(y, canon(moduli, fwd)), logn
        inv = np.stack([orc.poly_intt(moduli, a[i]) for i in range(3)])
        z = eng.to_host(eng.intt_residues_(moduli, eng.to_device(a)))
        assert np.array_equal(z, canon(moduli, inv)), logn
        eng.sync()   # in range: the guard stays quiet


def test_range_guard_flags_words_that_are_not_lazy(enga):
    """level A converts words with an exact-below-2^52 bit trick and its growth bounds assume lazy words
    (< 2q); hehub's transforms take any u64 (ntt.cpp:155-175) and level B reproduces that.  A word outside the range must not
    pass silently: the kernels set a sticky flag and the next synchronising call returns HP_ERANGE -- once; then the flag is clear."""
    from hehub_amd import capi
    from hehub_amd.engine import HpError

    eng = enga
    logn, mext = 12, [P.P50[1]] + P.P40[:3] + [P.P50[0]]
    n, L = 1 << logn, len(mext) - 1
    q = mext[:L]
    rng = SplitMix(4242)
    good = rng.poly((2, 2, L, n), q)
    dkey = eng.to_device(rng.poly((L, 2, L + 1, n), mext))

    def expect_flag(run):
        eng.sync()                      # clean before
        run()
        with pytest.raises(HpError) as e:
            eng.sync()
        assert e.value.code == capi.HP_ERANGE and "lazy word" in e.value.msg
        eng.sync()                      # reported once, then clear

    for bad_word in (1 << 52, (1 << 52) + 12345, (1 << 63) + 7, 2 * q[1] + (1 << 33)):
        for entry in ("ntt_residues", "intt_residues", "rescale", "rescale_kept_row", "relin_addend"):
            x = good.copy()
            limb = 1                    # a 40-bit limb: 2q + 2^33 is far below 2^52 and still not a lazy word
            if entry == "rescale":
                x[1, 0, L - 1, 77] = U(bad_word)          # the limb that is dropped: read by the level-A inverse transform
            else:
                x[1, 1, limb, n - 3] = U(bad_word)
            if entry == "ntt_residues":
                expect_flag(lambda: eng.ntt_residues_(q, eng.to_device(x[:, 1])))
            elif entry == "intt_residues":
                expect_flag(lambda: eng.intt_residues_(q, eng.to_device(x[:, 1])))
            elif entry == "relin_addend":   # quad[0] / quad[1] are the addend rows of the drop that ends relinearize (ckks/arith.cpp:70-71)
                quad = np.concatenate([x, good[:, :1]], axis=1)
                expect_flag(lambda: eng.ckks_relinearize(mext, eng.to_device(quad), dkey))
            else:
                expect_flag(lambda: eng.ckks_rescale(q, eng.to_device(x)))   # x rows of the drop epilogue / its inverse launch
    # words in range: nothing is flagged, whatever ran before
    eng.ckks_rescale(q, eng.to_device(good))
    eng.ntt_residues_(q, eng.to_device(good[:, 0]))
    eng.sync()
    # level B takes any word: no flag can come from there
    eng.set_parity_level("B")
    try:
        x = good.copy()
        x[0, 0, 0, 0] = U((1 << 63) + 9)
        eng.ckks_rescale(q, eng.to_device(x))
        eng.sync()
    finally:
        eng.set_parity_level("A")


@pytest.mark.gpu
def test_range_guard_is_kept_per_lane(enga):
    """the sticky range word belongs to the family, but `a level-A call of mine has not been checked` is kept per context:
    a lane that synchronises first must not use up the report of a lane whose bad call it did not wait for -- every lane that had
    level-A work pending when the word tripped reports HP_ERANGE once, a lane without such work reports nothing."""
    from hehub_amd import capi
    from hehub_amd.engine import HpError

    root = enga
    logn, mext = 12, [P.P50[1]] + P.P40[:3] + [P.P50[0]]
    n, L = 1 << logn, len(mext) - 1
    q = mext[:L]
    rng = SplitMix(777)
    good = rng.poly((2, 2, L, n), q)
    bad = good.copy()
    bad[1, 1, 1, 5] = U(2 * q[1] + (1 << 33))
    lane1, lane2 = root.fork(), root.fork()
    for lane in (lane1, lane2):
        lane.set_parity_level("A")
    try:
        root.sync(); lane1.sync(); lane2.sync()
        d_good, d_bad, d_good2 = root.to_device(good), root.to_device(bad), root.to_device(good)
        root.torch.cuda.synchronize()
        root.ckks_rescale(q, d_good)          # level-A work on the root lane, in range
        out_bad = lane1.ckks_rescale(q, d_bad)   # the bad call runs on lane 1
        root.torch.cuda.synchronize()         # (everything has run: whoever looks first sees the word set)
        with pytest.raises(HpError) as e:     # the root had level-A work pending when the word tripped: it reports (conservative) ...
            root.sync()
        assert e.value.code == capi.HP_ERANGE
        with pytest.raises(HpError) as e:     # ... and that did NOT use up lane 1's own report (round 5: lane 1 returned HP_OK here)
            lane1.sync()
        assert e.value.code == capi.HP_ERANGE
        lane2.sync()                          # a lane without level-A work pending reports nothing
        root.sync(); lane1.sync()             # reported once each, then clear
        lane2.ckks_rescale(q, d_good2)        # work that starts after the trip was counted is not blamed for it
        lane2.sync()
        del out_bad
    finally:
        lane1.close(); lane2.close()
