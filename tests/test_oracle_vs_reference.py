"""Pins the oracle: the C restatement (oracle/hehub_oracle.c) must agree
word-for-word (raw lazy u64 words, "Level B" of SURVEY.md section 8) with the
unmodified reference compiled into oracle/_ref/libhehub_ref.so.

Runs only where the reference was built (this container); skipped on the GPU
box, where the committed golden vectors (test_oracle_golden.py) take over.
"""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

U = np.uint64


def lcg_words(n, seed=1):
    # deterministic any-u64 inputs (full 64-bit range incl. values >= 2q)
    return SplitMix(seed).words(n, 0)


def test_scalar_helpers(orc, ref):
    rng = SplitMix(7)
    for q in P.NTT_TEST_Q + P.P40[:3] + P.P50[:2]:
        w = rng.words(64, 0)
        for a, b in zip(w[:32], w[32:]):
            b = int(b) % q
            bh = (b << 64) // q
            assert orc.mul_mod_harvey_lazy(q, int(a), b, bh) == ref.mul_mod_harvey_lazy(q, int(a), b, bh)
        for e in [1, 2, 65537, q - 1, int(w[0]) % q or 1, P.P50[3]]:
            if e % q == 0:
                continue
            assert orc.inverse_mod_prime(e, q) == ref.inverse_mod_prime(e, q)
        for idx in [0, 1, 2, 5, 1 << 20, q - 2, 2 * 4096 - 1]:
            assert orc.pow_mod(q, 3, idx) == ref.pow_mod(q, 3, idx)
    for bits in range(1, 17):
        for x in [0, 1, (1 << bits) - 1, (0x5A5A5A5A & ((1 << bits) - 1))]:
            assert orc.bit_rev(x, bits) == ref.bit_rev(x, bits)
    for q in P.NTT_TEST_Q:
        for logn in (4, 7, 12, 15):
            if (q - 1) % (2 << logn) == 0:
                assert orc.unity_root(q, 1 << logn) == ref.unity_root(q, 1 << logn)
    with pytest.raises(ValueError):
        orc.unity_root(260898817, 1 << 15 << 8)
    with pytest.raises(ValueError):
        ref.unity_root(260898817, 1 << 15 << 8)


@pytest.mark.parametrize("q", P.BARRETT_TEST_Q + P.P40[:2] + P.P50[:1] + [P.C1_Q])
def test_batched_mod_arith(orc, ref, q):
    n = 4096
    a = lcg_words(n, 11)
    b = lcg_words(n, 12)
    assert (orc.batched_barrett_lazy(q, a) == ref.batched_barrett_lazy(q, a)).all()
    assert (orc.batched_barrett(q, a) == ref.batched_barrett(q, a)).all()
    assert (orc.batched_reduce_strict(q, a) == ref.batched_reduce_strict(q, a)).all()
    if q % 2 == 1:
        # lazy inputs < 2q and fully arbitrary inputs
        for aa, bb in ((a % U(2 * q), b % U(2 * q)), (a % U(q), b % U(q)), (a, b)):
            assert (orc.mul_hybrid_lazy(q, aa, bb) == ref.mul_hybrid_lazy(q, aa, bb)).all()
        in128 = np.stack([a, b % U(q)], axis=1)
        assert (orc.montgomery_128_lazy(q, in128) == ref.montgomery_128_lazy(q, in128)).all()
    assert (orc.mul_barrett_lazy(q, a % U(q), b % U(q)) == ref.mul_barrett_lazy(q, a % U(q), b % U(q))).all()


@pytest.mark.parametrize("logn", [1, 2, 4, 7, 10, 12, 13, 14, 15])
@pytest.mark.parametrize("q", P.NTT_TEST_Q + [P.P50[0], P.P40[0]])
def test_ntt_raw_words(orc, ref, logn, q):
    if (q - 1) % (2 << logn) != 0:
        with pytest.raises(ValueError):
            orc.ntt(logn, q, np.zeros(1 << logn, dtype=U))
        with pytest.raises(ValueError):
            ref.ntt(logn, q, np.zeros(1 << logn, dtype=U))
        return
    n = 1 << logn
    rng = SplitMix(100 + logn)
    for x in (rng.words(n, q), rng.words(n, 2 * q), np.zeros(n, dtype=U), np.full(n, q - 1, dtype=U)):
        y_o, y_r = orc.ntt(logn, q, x), ref.ntt(logn, q, x)
        assert (y_o == y_r).all()
        z_o, z_r = orc.intt(logn, q, y_o), ref.intt(logn, q, y_r)
        assert (z_o == z_r).all()
        assert (orc.batched_reduce_strict(q, z_o) == x % U(q)).all()
        # inverse of a non-transformed (lazy) input as well
        assert (orc.intt(logn, q, x) == ref.intt(logn, q, x)).all()


def test_ntt_rejects_60_bit_modulus(orc, ref):
    x = np.zeros(16, dtype=U)
    with pytest.raises(ValueError):
        orc.ntt(4, 1152921504606844417, x)  # 2^60 - 2^14 + 1... any 60-bit value is refused first
    with pytest.raises(ValueError):
        ref.ntt(4, 1152921504606844417, x)


@pytest.mark.parametrize("n,moduli", [(8, [17179672577, 17179410433, 17176854529]),
                                       (1024, P.P40[:4]), (8192, P.C5_Q)])
def test_poly_operators(orc, ref, n, moduli):
    rng = SplitMix(n)
    L = len(moduli)
    two_q = [2 * m for m in moduli]
    a = rng.poly((L, n), two_q)
    b = rng.poly((L, n), two_q)
    assert (orc.poly_add(moduli, a, b) == ref.poly_add(moduli, a, b)).all()
    assert (orc.poly_sub(moduli, a, b) == ref.poly_sub(moduli, a, b)).all()
    assert (orc.poly_mul(moduli, a, b) == ref.poly_mul(moduli, a, b)).all()
    for s in (0, 1, 65537, moduli[0], 2**63 + 12345):
        assert (orc.poly_scalar_mul(moduli, a, s) == ref.poly_scalar_mul(moduli, a, s)).all()
    sc = [int(x) for x in rng.words(L, 0)]
    assert (orc.poly_rns_scalar_mul(moduli, a, sc) == ref.poly_rns_scalar_mul(moduli, a, sc)).all()
    assert (orc.poly_reduce_strict(moduli, a) == ref.poly_reduce_strict(moduli, a)).all()
    y = orc.poly_ntt(moduli, a)
    assert (y == ref.poly_ntt(moduli, a)).all()
    assert (orc.poly_intt(moduli, y) == ref.poly_intt(moduli, y)).all()
    assert (orc.poly_involution(a) == ref.poly_involution(a)).all()
    for step in (0, 1, 2, 3, n // 4 + 1):
        assert (orc.poly_cycle(a, step) == ref.poly_cycle(a, step)).all()


SCHEME_CASES = [
    # (logn, moduli_ext (q_0..q_{L-1}, p))
    (3, [1099510054913, 1073479681, 1072496641, 1099507695617]),   # tests/ckks_t.cpp-like {40,30,30}+40
    (10, P.P40[:3] + [P.P50[0]]),
    (12, [P.P50[1]] + P.P40[:4] + [P.P50[0]]),
]


@pytest.mark.parametrize("logn,mext", SCHEME_CASES)
def test_key_switch_and_scheme_level(orc, ref, logn, mext):
    n = 1 << logn
    L = len(mext) - 1
    rng = SplitMix(1000 + logn)
    q = mext[:L]
    ct1 = rng.poly((2, L, n), q)
    ct2 = rng.poly((2, L, n), q)
    key = rng.poly((L, 2, L + 1, n), mext)
    quad_o = orc.mult_low_level(q, ct1, ct2)
    assert (quad_o == ref.mult_low_level(q, ct1, ct2)).all()
    ext_o = orc.ext_prod(mext, quad_o[2], key)
    assert (ext_o == ref.ext_prod(mext, quad_o[2], key)).all()
    # drop p from the (L+1)-limb key-switch output, and q_last from an L-limb ct
    assert (orc.ckks_rescale(mext, ext_o) == ref.ckks_rescale(mext, ext_o)).all()
    assert (orc.ckks_rescale(q, ct1) == ref.ckks_rescale(q, ct1)).all()
    for t in (65537, 2, 1):
        assert (orc.bgv_mod_drop(mext, t, ext_o) == ref.bgv_mod_drop(mext, t, ext_o)).all()
        assert (orc.bgv_mod_drop(q, t, ct2) == ref.bgv_mod_drop(q, t, ct2)).all()
    lin_o = orc.ckks_relinearize(mext, quad_o, key)
    assert (lin_o == ref.ckks_relinearize(mext, quad_o, key)).all()
    assert (orc.bgv_relinearize(mext, quad_o, key) == ref.bgv_relinearize(mext, quad_o, key)).all()
    for step in (1, 2, 5):
        assert (orc.ckks_rotate(mext, ct1, key, step) == ref.ckks_rotate(mext, ct1, key, step)).all()
    assert (orc.ckks_conjugate(mext, ct2, key) == ref.ckks_conjugate(mext, ct2, key)).all()
    assert (orc.ckks_mult(mext, ct1, ct2, key) == ref.ckks_mult(mext, ct1, ct2, key)).all()
    assert (orc.bgv_mult(mext, 65537, ct1, ct2, key) == ref.bgv_mult(mext, 65537, ct1, ct2, key)).all()


@pytest.mark.parametrize("logn,mext", SCHEME_CASES[:2])
def test_key_switch_with_arbitrary_words(orc, ref, logn, mext):
    """ext_prod_montgomery never looks at the size of a word (wrapping u64 / u128 throughout): the restatement must agree
    with the reference on words that use all 64 bits too -- the GPU's carry paths are pinned against exactly this."""
    import numpy as np

    n, L = 1 << logn, len(mext) - 1
    rng = SplitMix(4242 + logn)
    pt = rng.words(L * n).reshape(L, n)
    key = rng.words(L * 2 * (L + 1) * n).reshape(L, 2, L + 1, n)
    key[:, :, :, ::3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    pt[:, 1::3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    assert (orc.ext_prod(mext, pt, key) == ref.ext_prod(mext, pt, key)).all()


def test_bgv_relinearize_quirk(orc, ref):
    """the reference's bgv::relinearize returns (quad[0], quad[1])
    as residues because its inner mod switch runs with plain_modulus == 1."""
    logn, mext = 10, P.P40[:3] + [P.P50[0]]
    n, L = 1 << logn, 3
    rng = SplitMix(5)
    quad = rng.poly((3, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    out = ref.bgv_relinearize(mext, quad, key)
    for h in range(2):
        assert (ref.poly_reduce_strict(mext[:L], out[h]) == quad[h]).all()
    assert (orc.bgv_relinearize(mext, quad, key) == out).all()


@pytest.mark.slow
def test_c3_shape_single_ct(orc, ref):
    """One ciphertext pair at the C3 shape (N=32768, L=10): raw words identical."""
    logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
    n, L = 1 << logn, 10
    rng = SplitMix(3)
    ct1 = rng.poly((2, L, n), mext[:L])
    ct2 = rng.poly((2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    assert (orc.ckks_mult(mext, ct1, ct2, key) == ref.ckks_mult(mext, ct1, ct2, key)).all()


@pytest.mark.parametrize("logn,moduli", [(3, P.P40[:2]), (8, [P.P50[1]] + P.P40[:2]), (12, P.P40[:4]), (13, P.C5_Q)])
def test_encrypt_decrypt_cores(orc, ref, logn, moduli):
    rng = SplitMix(500 + logn)
    noise, c1, pt, sk = P.edge_case(rng, logn, moduli)
    ct_o, ct_r = orc.rlwe_encrypt_core(moduli, noise, c1, pt, sk), ref.rlwe_encrypt_core(moduli, noise, c1, pt, sk)
    assert (ct_o == ct_r).all()
    back_o, back_r = orc.rlwe_decrypt_core(moduli, ct_o, sk), ref.rlwe_decrypt_core(moduli, ct_o, sk)
    assert (back_o == back_r).all()
    # decrypt(encrypt(pt)) = pt + e: the noise comes back as its lift
    n = 1 << logn
    for k, q in enumerate(moduli):
        exp = (pt[k].astype(object) + noise.astype(object)) % q
        assert (back_o[k].astype(object) == exp).all()


@pytest.mark.parametrize("n", [8, 4096])
def test_rns_base_transforms(orc, ref, n):
    rng = SplitMix(600 + n)
    for old, new in ((65537, P.P40[:3]), (P.P50[0], P.P40[:3]), (P.P40[0], [65537, P.P50[1]])):
        x = rng.words(n, 2 * old)                      # lazy input: rns_transform.cpp:113 reduces it strictly first
        assert (orc.rns_base_from_single(old, new, x) == ref.rns_base_from_single(old, new, x)).all()
    for old, new in ((P.P40[:3], 65537), ([P.P50[0], P.P40[1]], 65537), (P.P40[:2], P.P50[2])):
        x = P.small_rns_poly(rng, n, old)
        x[0, 0] += np.uint64(old[0])                   # one lazy word
        ok, got = orc.rns_base_to_single_small(old, new, x)
        assert ok and (got == ref.rns_base_to_single_small(old, new, x)[1]).all()
        big = rng.poly((len(old), n), old)             # uniform limbs are not a small polynomial: CRT branch
        assert not orc.rns_base_to_single_small(old, new, big)[0]


@pytest.mark.parametrize("n", [8, 512])
def test_rns_base_to_single_crt_branch(orc, ref, n):
    """rns_transform.cpp:86-104: coefficients that are not small go through the reference's big-integer CRT."""
    rng = SplitMix(650 + n)
    for old, new in ((P.P40[:3], 65537), ([P.P50[0], P.P40[1]], 65537), (P.P40[:5], 257), (P.P40[:2], P.P50[2]), (P.C3_Q, 786433)):
        x = rng.poly((len(old), n), old)
        x[0, 0] += np.uint64(old[0])                               # one lazy word
        assert (orc.rns_base_to_single(old, new, x) == ref.rns_base_to_single(old, new, x)).all()
        # values just around Q/2 and the "multiple of the new modulus" corner: x = -new (mod Q) gives new itself, not 0
        y = np.stack([np.full(n, (q - new % q) % q, dtype=np.uint64) for q in old])
        y[:, 1:] = x[:, 1:]
        got = orc.rns_base_to_single(old, new, y)
        assert (got == ref.rns_base_to_single(old, new, y)).all() and got[0] == new
        small = P.small_rns_poly(rng, n, old)
        assert (orc.rns_base_to_single(old, new, small) == ref.rns_base_to_single(old, new, small)).all()
