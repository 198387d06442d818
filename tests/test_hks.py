"""Hybrid key switch (extension; include/hehub_amd.h, hehub_amd/csrc/hp_hks.hip).

hehub has nothing to compare with, so the device path is pinned two ways:
  * word for word against an EXACT model written here with Python integers for the ModUp / ModDown compositions and the
    oracle's own transform / Montgomery primitives for everything hehub also does (so lazy representations agree);
  * semantically: with keys generated here the switched ciphertext decrypts to pt * s_from up to a noise far below the
    modulus, and a multiplication + relinearisation + rescale decrypts to the product."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu
U = np.uint64


@pytest.fixture(scope="module")
def eng():
    from hehub_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def crt(residues, moduli):
    Q = 1
    for q in moduli:
        Q *= q
    x = 0
    for r, q in zip(residues, moduli):
        M = Q // q
        x += int(r) * M * pow(M, -1, q)
    return x % Q, Q


def digits_of(L, alpha):
    return [list(range(d * alpha, min((d + 1) * alpha, L))) for d in range((L + alpha - 1) // alpha)]


def model_switch(orc, logn, mext, L, k, alpha, pt, key):
    """exact model of hp_dev_hks_switch for one polynomial: pt [L][n], key [nd][2][E][n] -> [2][L][n]"""
    n, E = 1 << logn, L + k
    digs = digits_of(L, alpha)
    coef = orc.poly_reduce_strict(mext[:L], orc.poly_intt(mext[:L], pt))
    D = np.zeros((len(digs), E, n), dtype=U)
    for d, limbs in enumerate(digs):
        ints = [crt([coef[a][i] for a in limbs], [mext[a] for a in limbs])[0] for i in range(n)]
        for m in range(E):
            if m in limbs:
                D[d, m] = pt[m]
            else:
                lifted = np.array([x % mext[m] for x in ints], dtype=U)
                D[d, m] = orc.ntt(logn, mext[m], lifted)
    ks = np.zeros((2, E, n), dtype=U)
    for h in range(2):
        for m in range(E):
            acc = [sum(int(D[d, m, i]) * int(key[d, h, m, i]) for d in range(len(digs))) for i in range(n)]
            pairs = np.array([[a & (2**64 - 1), a >> 64] for a in acc], dtype=U)
            ks[h, m] = orc.montgomery_128_lazy(mext[m], pairs)
    pm = mext[L:]
    out = np.zeros((2, L, n), dtype=U)
    for h in range(2):
        yp = orc.poly_reduce_strict(pm, orc.poly_intt(pm, np.ascontiguousarray(ks[h, L:])))
        Ys = [crt([yp[j][i] for j in range(k)], pm) for i in range(n)]
        rem = np.zeros((L, n), dtype=U)
        for i_q in range(L):
            q = mext[i_q]
            vals = [(y % q) if y < Pm // 2 else q - ((Pm - y) % q) for (y, Pm) in Ys]
            rem[i_q] = orc.ntt(logn, q, np.array(vals, dtype=U))
        diff = orc.poly_sub(mext[:L], np.ascontiguousarray(ks[h, :L]), rem)
        Pprod = 1
        for p in pm:
            Pprod *= p
        out[h] = orc.poly_rns_scalar_mul(mext[:L], diff, [pow(Pprod % q, -1, q) for q in mext[:L]])
    return out


def keygen(orc, rng, logn, mext, L, k, alpha, s_to, s_from, sigma_bound=8):
    """hybrid key u64[nd][2][E][n]: row d = RLWE encryption under s_to of (P mod q_i) * s_from on the limbs of digit d."""
    n, E = 1 << logn, L + k
    digs = digits_of(L, alpha)
    Pprod = 1
    for p in mext[L:]:
        Pprod *= p
    key = np.zeros((len(digs), 2, E, n), dtype=U)
    to_ntt = orc.poly_reduce_strict(mext, orc.poly_ntt(mext, np.stack([(s_to % q).astype(U) for q in mext])))
    from_ntt = orc.poly_reduce_strict(mext, orc.poly_ntt(mext, np.stack([(s_from % q).astype(U) for q in mext])))
    for d, limbs in enumerate(digs):
        a = rng.poly((E, n), mext)                                  # uniform, NTT form
        e = (rng.words(n, 2 * sigma_bound + 1).astype(np.int64) - sigma_bound)
        e_ntt = orc.poly_reduce_strict(mext, orc.poly_ntt(mext, np.stack([(e % q).astype(U) for q in mext])))
        b = orc.poly_sub(mext, e_ntt, orc.poly_mul(mext, a, to_ntt))          # e - a*s
        msg = np.zeros((E, n), dtype=U)
        for m in limbs:
            msg[m] = orc.poly_rns_scalar_mul([mext[m]], from_ntt[m][None], [Pprod % mext[m]])[0]
        b = orc.poly_reduce_strict(mext, orc.poly_add(mext, b, msg))
        mont = [(1 << 64) % q for q in mext]                        # Montgomery form, as hehub stores its keys
        key[d, 0] = orc.poly_reduce_strict(mext, orc.poly_rns_scalar_mul(mext, b, mont))
        key[d, 1] = orc.poly_reduce_strict(mext, orc.poly_rns_scalar_mul(mext, a, mont))
    return key


def centred_error(orc, logn, moduli, poly_ntt):
    """max |coefficient| of an NTT-form RNS polynomial, read as a centred integer"""
    c = orc.poly_reduce_strict(moduli, orc.poly_intt(moduli, poly_ntt))
    worst = 0
    for i in range(c.shape[1]):
        x, Q = crt([c[a][i] for a in range(len(moduli))], moduli)
        worst = max(worst, min(x, Q - x))
    return worst


@pytest.mark.parametrize("logn,L,k,alpha", [(4, 4, 2, 2), (5, 5, 2, 2), (11, 3, 1, 1), (6, 6, 3, 3), (5, 4, 4, 4), (4, 3, 9, 3), (11, 2, 9, 1),
                                           (5, 8, 5, 8), (4, 5, 8, 2), (15, 3, 2, 2)])
def test_switch_matches_the_exact_model(eng, orc, logn, L, k, alpha):
    mext = P.P40[:L] + (P.P50 + P.P40[L:])[:k]
    n = 1 << logn
    rng = SplitMix(1100 + logn + L)
    B = 2
    nd = (L + alpha - 1) // alpha
    pt = np.stack([rng.poly((L, n), mext[:L]) for _ in range(B)])
    key = rng.poly((nd, 2, L + k, n), mext)                          # any words: the model is about arithmetic, not security
    got = eng.to_host(eng.hks_switch(mext, k, alpha, eng.to_device(pt), eng.to_device(key)))
    for i in range(B):
        assert np.array_equal(got[i], model_switch(orc, logn, mext, L, k, alpha, pt[i], key)), (i, logn, L, k, alpha)


@pytest.mark.parametrize("logn,L,k,alpha", [(5, 4, 2, 2), (6, 6, 2, 2), (6, 6, 3, 3), (11, 4, 2, 2)])
def test_switch_decrypts_to_pt_times_s_from(eng, orc, logn, L, k, alpha):
    mext = P.P40[:L] + P.P50[:k]                                     # 50-bit special primes: P >= every 2..3 x 40-bit digit
    n = 1 << logn
    rng = SplitMix(1200 + logn + L)
    s = rng.words(n, 3).astype(np.int64) - 1                         # ternary secret
    s2 = None
    q = mext[:L]
    s_ntt = orc.poly_reduce_strict(q, orc.poly_ntt(q, np.stack([(s % m).astype(U) for m in q])))
    s2_ntt = orc.poly_mul(q, s_ntt, s_ntt)
    s2_coef = orc.poly_reduce_strict(q, orc.poly_intt(q, s2_ntt))
    s2 = np.array([(lambda x, Q: x if x < Q // 2 else x - Q)(*crt([s2_coef[a][i] for a in range(L)], q)) for i in range(n)], dtype=object)
    key = keygen(orc, rng, logn, mext, L, k, alpha, s, np.array([int(v) for v in s2], dtype=object))
    pt = rng.poly((L, n), q)
    out = eng.to_host(eng.hks_switch(mext, k, alpha, eng.to_device(pt[None]), eng.to_device(key)))[0]
    # out0 + out1*s - pt*s^2 must be small
    lhs = orc.poly_add(q, out[0], orc.poly_mul(q, out[1], s_ntt))
    err = orc.poly_sub(q, lhs, orc.poly_mul(q, pt, s2_ntt))
    worst = centred_error(orc, logn, q, err)
    Q = 1
    for m in q:
        Q *= m
    assert worst < (1 << 24) and worst * (1 << 60) < Q, worst      # noise ~ dnum * N * sigma, nowhere near Q


@pytest.mark.parametrize("logn,L,k,alpha", [(5, 4, 2, 2), (4, 2, 1, 1), (11, 4, 2, 2), (11, 2, 3, 2), (12, 5, 1, 3), (13, 3, 2, 1)])
def test_mult_relin_rescale_with_a_hybrid_key(eng, orc, logn, L, k, alpha):
    """ckks::mult_low_level + hybrid relinearisation + rescale equals the composition of the pieces (tensor and rescale
    are hehub's, the switch is the model's).  The tiled sizes merge ModDown and the rescale into one transform per limb,
    which yields another lazy representative of the same residues: the comparison is on strict residues, and the
    two-step composition (HP_HKS_TWO_STEP) is held to the words themselves."""
    mext = P.P40[:L] + P.P50[:k]
    n, q = 1 << logn, mext[:L]
    rng = SplitMix(1300 + logn + L)
    nd = (L + alpha - 1) // alpha
    ct1 = rng.poly((2, L, n), q); ct2 = rng.poly((2, L, n), q)
    key = rng.poly((nd, 2, L + k, n), mext)
    got = eng.to_host(eng.ckks_mult_hks(mext, k, alpha, eng.to_device(ct1[None]), eng.to_device(ct2[None]), eng.to_device(key)))[0]
    quad = orc.mult_low_level(q, ct1, ct2)
    sw = model_switch(orc, logn, mext, L, k, alpha, quad[2], key)
    lin = np.stack([orc.poly_add(q, sw[0], quad[0]), orc.poly_add(q, sw[1], quad[1])])
    exp = orc.ckks_rescale(q, lin)
    assert got.shape == exp.shape
    assert (got < 2 * np.array(q[:-1], dtype=U)[None, :, None]).all()
    for h in range(2):
        assert np.array_equal(orc.poly_reduce_strict(q[:-1], got[h]), orc.poly_reduce_strict(q[:-1], exp[h])), h
    if logn < 11:
        assert np.array_equal(got, exp)


def test_two_step_hybrid_mult_is_word_exact(orc, monkeypatch):
    """HP_HKS_TWO_STEP: switch, then hehub's rescale as a separate step: the model's words exactly, at a tiled size too"""
    from hehub_amd.engine import Engine

    monkeypatch.setenv("HP_HKS_TWO_STEP", "1")
    e = Engine(0)
    try:
        logn, L, k, alpha = 11, 3, 2, 2
        mext = P.P40[:L] + P.P50[:k]
        n, q = 1 << logn, mext[:L]
        rng = SplitMix(1350)
        ct1 = rng.poly((2, L, n), q); ct2 = rng.poly((2, L, n), q)
        key = rng.poly(((L + alpha - 1) // alpha, 2, L + k, n), mext)
        got = e.to_host(e.ckks_mult_hks(mext, k, alpha, e.to_device(ct1[None]), e.to_device(ct2[None]), e.to_device(key)))[0]
        quad = orc.mult_low_level(q, ct1, ct2)
        sw = model_switch(orc, logn, mext, L, k, alpha, quad[2], key)
        lin = np.stack([orc.poly_add(q, sw[0], quad[0]), orc.poly_add(q, sw[1], quad[1])])
        assert np.array_equal(got, orc.ckks_rescale(q, lin))
    finally:
        e.close()


@pytest.mark.parametrize("logn,L,k,alpha", [(5, 4, 2, 2), (11, 3, 2, 2)])
def test_rotate_and_conjugate_with_a_hybrid_key(eng, orc, logn, L, k, alpha):
    """gather + hybrid switch of the moved c1 + the moved c0 added to polynomial 0 only (ckks/arith.cpp:75-93 with the
    hybrid switch in place of ext_prod + rescale)."""
    mext = P.P40[:L] + P.P50[:k]
    n, q = 1 << logn, mext[:L]
    rng = SplitMix(1400 + logn)
    nd = (L + alpha - 1) // alpha
    ct = rng.poly((2, L, n), q)
    key = rng.poly((nd, 2, L + k, n), mext)
    d_ct, d_key = eng.to_device(ct[None]), eng.to_device(key)
    for conj in (False, True):
        moved = np.stack([orc.poly_involution(ct[h]) if conj else orc.poly_cycle(ct[h], 3) for h in range(2)])
        sw = model_switch(orc, logn, mext, L, k, alpha, moved[1], key)
        exp = np.stack([orc.poly_add(q, sw[0], moved[0]), sw[1]])
        got = eng.ckks_conjugate_hks(mext, k, alpha, d_ct, d_key) if conj else eng.ckks_rotate_hks(mext, k, alpha, d_ct, d_key, 3)
        assert np.array_equal(eng.to_host(got)[0], exp), conj


def test_merged_and_two_step_hybrid_mult_agree_on_random_shapes(eng, orc, monkeypatch):
    """The merged ModDown + rescale (one transform per limb) against the two-step composition -- itself held to the exact
    model word for word above -- on random tiled shapes: same strict residues everywhere, lazy words below 2q."""
    from hehub_amd.engine import Engine

    monkeypatch.setenv("HP_HKS_TWO_STEP", "1")
    two = Engine(0)
    monkeypatch.delenv("HP_HKS_TWO_STEP")
    try:
        rng = SplitMix(20260928)
        pool = P.P40 + P.P50
        for case in range(24):
            logn = 11 + int(rng.words(1, 3)[0])                       # 11..13
            L = 2 + int(rng.words(1, 9)[0])                           # 2..10
            k = 1 + int(rng.words(1, 4)[0])                           # 1..4
            alpha = 1 + int(rng.words(1, min(L, 8))[0])
            B = 1 + int(rng.words(1, 3)[0])
            order = np.argsort(rng.words(len(pool)))                   # a random chain of distinct primes
            mext = [pool[i] for i in order[:L + k]]
            n, q = 1 << logn, mext[:L]
            nd = (L + alpha - 1) // alpha
            ct1 = np.stack([rng.poly((2, L, n), q) for _ in range(B)]); ct2 = np.stack([rng.poly((2, L, n), q) for _ in range(B)])
            key = rng.poly((nd, 2, L + k, n), mext)
            a = eng.to_host(eng.ckks_mult_hks(mext, k, alpha, eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)))
            b = two.to_host(two.ckks_mult_hks(mext, k, alpha, two.to_device(ct1), two.to_device(ct2), two.to_device(key)))
            qa = np.array(q[:-1], dtype=U)[None, None, :, None]
            assert (a < 2 * qa).all() and (b < 2 * qa).all(), (case, logn, L, k, alpha)
            assert np.array_equal(a % qa, b % qa), (case, logn, L, k, alpha, B)
    finally:
        two.close()


@pytest.mark.parametrize("logn,L,k,alpha", [(11, 4, 2, 2), (12, 5, 1, 3), (13, 6, 3, 3), (15, 3, 2, 2), (11, 2, 9, 1)])
def test_hybrid_key_switch_at_parity_level_a(eng, orc, logn, L, k, alpha):
    """Round 5: with hp_ctx_set_parity_level(HP_PARITY_A) the hybrid key switch runs its
    transforms -- the coefficient rows, the lifted digits (HP_NTT_HKS), the P-part inverse -- on the FP64 residue kernels.  Hybrid
    results have no word-level contract with hehub (other keys, rgsw.cpp:98-153 is the single-prime scheme); the contract is the
    exact integer model: same residues, every word a lazy word (< 2q), at both levels."""
    mext = P.P40[:L] + (P.P50 + P.P40[L:])[:k]
    n, q = 1 << logn, mext[:L]
    rng = SplitMix(1700 + logn + L)
    nd = (L + alpha - 1) // alpha
    B = 2
    pt = np.stack([rng.poly((L, n), q) for _ in range(B)])
    ct1 = rng.poly((B, 2, L, n), q); ct2 = rng.poly((B, 2, L, n), q)
    key = rng.poly((nd, 2, L + k, n), mext)
    dk = eng.to_device(key)
    qa = np.array(q, dtype=U)[None, :, None]
    res = {}
    for level in ("B", "A"):
        eng.set_parity_level(level)
        try:
            sw = eng.to_host(eng.hks_switch(mext, k, alpha, eng.to_device(pt), dk))
            rot = eng.to_host(eng.ckks_rotate_hks(mext, k, alpha, eng.to_device(ct1), dk, 3))
            mul = eng.to_host(eng.ckks_mult_hks(mext, k, alpha, eng.to_device(ct1), eng.to_device(ct2), dk)) if L >= 2 else None
            eng.sync()   # (level A: the range guard stays quiet on canonical rows)
        finally:
            eng.set_parity_level("B")
        res[level] = (sw, rot, mul)
    for i in range(B):
        model = model_switch(orc, logn, mext, L, k, alpha, pt[i], key)
        for level in ("B", "A"):
            got = res[level][0][i]
            assert (got < 2 * qa).all(), level
            assert np.array_equal(got % qa, model % qa), (level, i)
    for a, b, mods in ((res["B"][1], res["A"][1], q), (res["B"][2], res["A"][2], q[:-1])):
        if a is None:
            continue
        m = np.array(mods, dtype=U)[None, None, :, None]
        assert (b < 2 * m).all() and np.array_equal(a % m, b % m)
