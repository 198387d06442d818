"""Extensions beyond the reference (include/hehub_amd.h): a key generated at a higher level used
at a lower one, and rescale by several primes.  hehub throws in both cases, so they are pinned by equivalence: the
result must equal, word for word, what the reference-parity entry points (and the oracle) give for the extracted
sub-key / for successive single drops."""
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


@pytest.mark.parametrize("logn,L0,L", [(5, 4, 2), (11, 5, 3), (12, 4, 4), (13, 6, 5)])
def test_key_of_a_higher_level(eng, orc, logn, L0, L):
    n, B = 1 << logn, 2
    q_full, p = P.P40[:L0], P.P50[0]
    rng = SplitMix(900 + logn)
    key_full = rng.poly((L0, 2, L0 + 1, n), q_full + [p])
    sub = np.ascontiguousarray(key_full[:L][:, :, list(range(L)) + [L0], :])     # rows j < L, columns q_0..q_{L-1}, p
    mext = q_full[:L] + [p]
    ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    d1, d2, dk, dsub = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key_full), eng.to_device(sub)
    if L >= 2:
        exp = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], sub) for i in range(B)])
        assert np.array_equal(eng.to_host(eng.ckks_mult_at(mext, L0, d1, d2, dk)), exp)
        assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dsub)), exp)
    exp = np.stack([orc.ckks_rotate(mext, ct1[i], sub, 3) for i in range(B)])
    assert np.array_equal(eng.to_host(eng.ckks_rotate_at(mext, L0, d1, dk, 3)), exp)
    pt = np.ascontiguousarray(ct2[:, 1])
    exp = np.stack([orc.ext_prod(mext, pt[i], sub) for i in range(B)])
    assert np.array_equal(eng.to_host(eng.ext_prod_at(mext, L0, eng.to_device(pt), dk)), exp)
    from hehub_amd.engine import InvalidArgument

    with pytest.raises(InvalidArgument):
        eng.ckks_rotate_at(mext, L - 1, d1, dk, 1)          # a key for FEWER moduli than the ciphertext has is an error


@pytest.mark.parametrize("logn,L,drops", [(4, 3, 2), (11, 4, 2), (12, 5, 3), (13, 4, 1)])
def test_rescale_by_several_primes(eng, orc, logn, L, drops):
    n, B = 1 << logn, 2
    q = P.P40[:L]
    rng = SplitMix(950 + logn)
    ct = np.stack([rng.poly((2, L, n), q) for _ in range(B)])
    exp = ct
    for d in range(drops):
        exp = np.stack([orc.ckks_rescale(q[:L - d], exp[i]) for i in range(B)])
    assert np.array_equal(eng.to_host(eng.ckks_rescale_n(q, eng.to_device(ct), drops)), exp)
    from hehub_amd.engine import InvalidArgument

    with pytest.raises(InvalidArgument):
        eng.ckks_rescale_n(q, eng.to_device(ct), L)


@pytest.mark.parametrize("n", [8, 2048])
def test_many_to_many_base_conversion(eng, orc, n):
    """Pinned by equivalence: every output limb equals the reference-parity many -> one CRT composition into that modulus
    (a polynomial that is not small makes the oracle take that branch), and for exact integers in Python."""
    rng = SplitMix(990 + n)
    old, new = P.P40[:4], [65537, P.P50[1], P.P40[6], 257]
    x = np.stack([rng.poly((len(old), n), old) for _ in range(2)])
    got = eng.to_host(eng.rns_base_many_to_many(old, new, eng.to_device(x)))
    for i in range(2):
        for k, m in enumerate(new):
            assert np.array_equal(got[i, k], orc.rns_base_to_single(old, m, x[i]))
    # exact integers: CRT compose coefficient 0 of polynomial 0 with Python ints
    Q = 1
    for q in old:
        Q *= q
    big = sum(int(x[0, a, 0]) * (Q // q) * pow(Q // q, -1, q) for a, q in enumerate(old)) % Q
    for k, m in enumerate(new):
        assert int(got[0, k, 0]) == (big % m if big < Q // 2 else m - ((Q - big) % m))


def _negacyclic(a, b, t):
    """a * b in Z_t[X]/(X^N + 1), object-free: values stay below 2^63 for t < 2^24 and N <= 2^15"""
    n = len(a)
    full = np.convolve(a.astype(np.int64), b.astype(np.int64))
    lo, hi = full[:n].copy(), np.zeros(n, dtype=np.int64)
    hi[: n - 1] = full[n:]
    return (lo - hi) % t


@pytest.mark.parametrize("logn,L", [(6, 3), (11, 3), (12, 4)])
def test_bgv_product_decrypts_with_the_inner_plain_modulus(eng, orc, logn, L):
    """hp_dev_bgv_mult_relin_modswitch_t keeps the key-switched term that hehub's bgv::relinearize discards (its inner mod
    switch runs with plain_modulus 1, SURVEY.md 8c caveat 1).  Pinned two ways: word for word against the step-wise
    composition (oracle with inner_t = t), and by decryption: with a key made for it the result decrypts to m1 * m2."""
    n, t = 1 << logn, 65537
    q, p = P.P40[:L], P.P50[0]
    mext = q + [p]
    rng = SplitMix(1500 + logn)
    s = rng.words(n, 3).astype(np.int64) - 1

    def ntt_of(small, moduli):
        return orc.poly_reduce_strict(moduli, orc.poly_ntt(moduli, np.stack([(small % m).astype(U) for m in moduli])))

    s_q, s_ext = ntt_of(s, q), ntt_of(s, mext)
    s2_ext = orc.poly_reduce_strict(mext, orc.poly_mul(mext, s_ext, s_ext))

    def noise(moduli):
        return ntt_of(t * (rng.words(n, 17).astype(np.int64) - 8), moduli)

    def encrypt(m):
        a = rng.poly((L, n), q)
        b = orc.poly_add(q, orc.poly_sub(q, noise(q), orc.poly_mul(q, a, s_q)), ntt_of(m, q))
        return np.stack([orc.poly_reduce_strict(q, b), a])

    # relinearisation key: row j = (-a s + t e + (p mod q_j) ((p mod t)^-1 mod q_j) s^2 on limb j, a), Montgomery form
    key = np.zeros((L, 2, L + 1, n), dtype=U)
    mont = [(1 << 64) % m for m in mext]
    for j in range(L):
        a = rng.poly((L + 1, n), mext)
        b = orc.poly_sub(mext, noise(mext), orc.poly_mul(mext, a, s_ext))
        f = (p % q[j]) * pow(p % t, -1, q[j]) % q[j]
        msg = np.zeros((L + 1, n), dtype=U)
        msg[j] = orc.poly_rns_scalar_mul([q[j]], s2_ext[j][None], [f])[0]
        b = orc.poly_add(mext, b, msg)
        key[j, 0] = orc.poly_reduce_strict(mext, orc.poly_rns_scalar_mul(mext, b, mont))
        key[j, 1] = orc.poly_reduce_strict(mext, orc.poly_rns_scalar_mul(mext, a, mont))

    m1 = rng.words(n, t).astype(np.int64); m2 = rng.words(n, t).astype(np.int64)
    ct1, ct2 = encrypt(m1), encrypt(m2)
    d1, d2, dk = eng.to_device(ct1[None]), eng.to_device(ct2[None]), eng.to_device(key)
    got = eng.to_host(eng.bgv_mult(mext, t, d1, d2, dk, inner_t=True))[0]

    # (1) the step-wise composition, word for word
    quad = orc.mult_low_level(q, ct1, ct2)
    exp = orc.bgv_mod_drop(q, t, orc.bgv_relinearize(mext, quad, key, inner_t=t))
    assert np.array_equal(got, exp)
    step = eng.bgv_mod_switch(q, t, eng.bgv_relinearize(mext, eng.mult_low_level(q, d1, d2), dk, inner_t=t))
    assert np.array_equal(eng.to_host(step)[0], exp)

    # (2) decryption at the lower level: centred (c0 + c1 s) mod Q, then mod t
    ql = q[:-1]
    lhs = orc.poly_add(ql, got[0], orc.poly_mul(ql, got[1], s_q[:-1]))
    coef = orc.poly_reduce_strict(ql, orc.poly_intt(ql, lhs))
    Q = 1
    for m in ql:
        Q *= m
    inv = [(Q // m) * pow(Q // m, -1, m) for m in ql]
    dec = np.empty(n, dtype=np.int64)
    worst = 0
    for i in range(n):
        x = sum(int(coef[a][i]) * inv[a] for a in range(len(ql))) % Q
        x = x if x < Q // 2 else x - Q
        worst = max(worst, abs(x))
        dec[i] = x % t
    assert worst * (1 << 20) < Q, worst                                   # noise far below the modulus
    assert np.array_equal(dec, _negacyclic(m1, m2, t))

    # hehub's own composition (inner plain modulus 1) loses the term: same inputs, wrong plaintext
    ref = eng.to_host(eng.bgv_mult(mext, t, d1, d2, dk))[0]
    assert not np.array_equal(ref, got)
