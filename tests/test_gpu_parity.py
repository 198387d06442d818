"""GPU parity: every C-ABI entry point of the HIP engine against the oracle on identical seeded
inputs, bit-exact on raw lazy u64 words (Level B of SURVEY.md section 8).  Run with -m gpu."""
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


def eq(a, b):
    return a.shape == b.shape and bool((a == b).all())


# ---- drop-in host entry points (mirror tests/mod_arith_t.cpp, tests/ntt_t.cpp) ----------------
@pytest.mark.parametrize("q", P.BARRETT_TEST_Q + [P.P40[0], P.P50[0], P.C1_Q])
def test_host_mod_arith(eng, orc, q):
    for n in (1, 7, 4096, 100003):
        a = SplitMix(11 + n).words(n, 0)
        b = SplitMix(12 + n).words(n, 0)
        assert eq(eng.host_barrett_lazy(q, a), orc.batched_barrett_lazy(q, a))
        assert eq(eng.host_barrett(q, a), orc.batched_barrett(q, a))
        assert eq(eng.host_reduce_strict(q, a), orc.batched_reduce_strict(q, a))
        assert eq(eng.host_mul_barrett_lazy(q, a % U(q), b % U(q)), orc.mul_barrett_lazy(q, a % U(q), b % U(q)))
        if q % 2:
            for aa, bb in ((a % U(2 * q), b % U(2 * q)), (a, b)):
                assert eq(eng.host_mul_hybrid_lazy(q, aa, bb), orc.mul_hybrid_lazy(q, aa, bb))
            in128 = np.stack([a, b % U(q)], axis=1)
            assert eq(eng.host_montgomery_128_lazy(q, in128), orc.montgomery_128_lazy(q, in128))
    assert eng.host_barrett(q, np.zeros(0, dtype=U)).size == 0


@pytest.mark.parametrize("logn", [1, 2, 4, 7, 10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("q", P.NTT_TEST_Q + [P.P50[0], P.P40[0]])
def test_host_ntt_round_trip(eng, orc, logn, q):
    from hehub_amd.engine import InvalidArgument

    n = 1 << logn
    if (q - 1) % (2 * n):
        with pytest.raises(InvalidArgument, match="2N doesn't divide"):
            eng.host_ntt(logn, q, np.zeros(n, dtype=U))
        return
    rng = SplitMix(100 + logn)
    for x in (rng.words(n, q), rng.words(n, 2 * q), np.zeros(n, dtype=U), np.full(n, q - 1, dtype=U)):
        y = eng.host_ntt(logn, q, x)
        assert eq(y, orc.ntt(logn, q, x))
        z = eng.host_intt(logn, q, y)
        assert eq(z, orc.intt(logn, q, y))
        assert eq(orc.batched_reduce_strict(q, z), x % U(q))
    # delta and X (tests/ntt_t.cpp "one", "just x")
    d = np.zeros(n, dtype=U); d[0] = 1
    assert (eng.host_ntt(logn, q, d) % U(q) == 1).all()


def test_host_ntt_rejects_60_bit(eng):
    from hehub_amd.engine import InvalidArgument

    with pytest.raises(InvalidArgument, match="bit size > 59"):
        eng.host_ntt(4, 1152921504606844417, np.zeros(16, dtype=U))


@pytest.mark.parametrize("logn", [11, 12, 13, 14, 15])
def test_tiled_kernels_match_simple_kernels(eng, orc, logn):
    """the register/LDS-tiled transforms against the one-stage-at-a-time kernels on the GPU itself"""
    q = P.P50[0]
    x = SplitMix(7).words(1 << logn, 2 * q)
    eng.force_generic(True)
    yg, zg = eng.host_ntt(logn, q, x), eng.host_intt(logn, q, x)
    eng.force_generic(False)
    assert eq(eng.host_ntt(logn, q, x), yg) and eq(eng.host_intt(logn, q, x), zg)
    assert eq(yg, orc.ntt(logn, q, x))


# ---- device batches ---------------------------------------------------------------------------
BATCH_CASES = [(3, [17179672577, 17179410433, 17176854529], 5), (10, P.P40[:4], 3), (12, P.P40[:3], 9),
               (13, P.C5_Q, 4), (14, P.C2_MODULI, 6), (15, P.C3_Q[:3], 5)]


@pytest.mark.parametrize("logn,moduli,B", BATCH_CASES)
def test_dev_batch_ops(eng, orc, logn, moduli, B):
    n, L = 1 << logn, len(moduli)
    rng = SplitMix(logn * 31 + B)
    two_q = [2 * m for m in moduli]
    a = rng.poly((B, L, n), two_q)
    b = rng.poly((B, L, n), two_q)
    da, db = eng.to_device(a), eng.to_device(b)
    exp = lambda f: np.stack([f(i) for i in range(B)])
    assert eq(eng.to_host(eng.poly_add(moduli, da, db)), exp(lambda i: orc.poly_add(moduli, a[i], b[i])))
    assert eq(eng.to_host(eng.poly_sub(moduli, da, db)), exp(lambda i: orc.poly_sub(moduli, a[i], b[i])))
    assert eq(eng.to_host(eng.poly_mul(moduli, da, db)), exp(lambda i: orc.poly_mul(moduli, a[i], b[i])))
    sc = [int(w) for w in rng.words(L, 0)]
    assert eq(eng.to_host(eng.poly_scalar_mul(moduli, da, sc)), exp(lambda i: orc.poly_rns_scalar_mul(moduli, a[i], sc)))
    assert eq(eng.to_host(eng.poly_scalar_mul(moduli, da, 2**63 + 5)), exp(lambda i: orc.poly_scalar_mul(moduli, a[i], 2**63 + 5)))
    assert eq(eng.to_host(eng.poly_involution(da)), exp(lambda i: orc.poly_involution(a[i])))
    for step in (1, 3, n // 4 + 1):
        assert eq(eng.to_host(eng.poly_cycle(da, step)), exp(lambda i: orc.poly_cycle(a[i], step)))
    y = eng.to_host(eng.ntt_(moduli, da.clone()))
    assert eq(y, exp(lambda i: orc.poly_ntt(moduli, a[i])))
    z = eng.to_host(eng.intt_(moduli, eng.to_device(y)))
    assert eq(z, exp(lambda i: orc.poly_intt(moduli, y[i])))
    zs = eng.to_host(eng.intt_(moduli, eng.to_device(y), strict=True))
    assert eq(zs, exp(lambda i: orc.poly_reduce_strict(moduli, orc.poly_intt(moduli, y[i]))))
    assert eq(zs, a % np.array(moduli, dtype=U)[None, :, None])
    assert eq(eng.to_host(eng.poly_reduce_strict_(moduli, da.clone())), exp(lambda i: orc.poly_reduce_strict(moduli, a[i])))


# a chain of += / -= as one pass (hp_dev_poly_fold_rows): the words of the single calls applied in order (rns.cpp:58-118), for chains
# shorter and longer than one launch's 32 terms, at ring degrees below and above one workgroup's chunk
@pytest.mark.parametrize("logn,moduli,terms", [(1, [P.P40[0]], 2), (4, P.P40[:2], 5), (10, [P.P50[0], P.P40[1]], 33), (12, P.P40[:3], 16),
                                               (13, P.P40[:2] + [P.P50[1]], 70), (12, P.P40[:1], 1)])
def test_dev_poly_fold_rows(eng, orc, logn, moduli, terms):
    n, L = 1 << logn, len(moduli)
    rng = SplitMix(logn * 97 + terms)
    two_q = [2 * m for m in moduli]
    polys = 2
    x = rng.poly((polys, terms, L, n), two_q)
    negate = [0] + [int(w) & 1 for w in rng.words(terms - 1, 0)] if terms > 1 else [1]
    # the terms lie scattered: views of a bigger tensor in another order
    pool = eng.to_device(np.ascontiguousarray(x.transpose(1, 0, 2, 3)))      # [terms][polys][L][n]
    chains = [[pool[j][p] for j in range(terms)] for p in range(polys)]
    got = eng.to_host(eng.poly_fold_rows(moduli, chains, negate))
    for p in range(polys):
        acc = x[p][0]
        for j in range(1, terms):
            acc = orc.poly_sub(moduli, acc, x[p][j]) if negate[j] else orc.poly_add(moduli, acc, x[p][j])
        assert eq(got[p], acc)
    # in place: the result over the first term
    out = eng.poly_fold_rows(moduli, chains, negate, out=None)
    first = [c[0].clone() for c in chains]
    packed = eng.empty((polys, L, n))
    for p in range(polys):
        packed[p].copy_(first[p])
    eng.poly_fold_rows(moduli, [[packed[p]] + chains[p][1:] for p in range(polys)], negate, out=packed)
    assert eq(eng.to_host(packed), eng.to_host(out))


SCHEME_CASES = [
    (3, [1099510054913, 1073479681, 1072496641, 1099507695617], 4),     # tests/ckks_t.cpp shape {40,30,30}+40, N=8
    (7, [P.P40[0], P.P40[1], P.P50[0]], 3),
    (11, P.P40[:3] + [P.P50[0]], 2),
    (12, [P.P50[1]] + P.P40[:4] + [P.P50[0]], 3),
    (P.C5_LOGN, P.C5_MODULI_EXT, 2),
    (14, P.P50[1:4] + [P.P50[0]], 2),     # C2 ring degree, fused drop-last-prime path
    (16, [P.P40[0], 1125899902124033, P.P50[1]], 1),   # N = 65536 (primes = 1 mod 2^17): the largest degree the reference takes
]


@pytest.mark.parametrize("logn,mext,B", SCHEME_CASES)
def test_dev_scheme_level(eng, orc, logn, mext, B):
    n, L = 1 << logn, len(mext) - 1
    q = mext[:L]
    rng = SplitMix(1000 + logn)
    ct1 = rng.poly((B, 2, L, n), q)
    ct2 = rng.poly((B, 2, L, n), q)
    key = rng.poly((L, 2, L + 1, n), mext)
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    exp = lambda f: np.stack([f(i) for i in range(B)])
    quad = eng.mult_low_level(q, d1, d2)
    quad_h = eng.to_host(quad)
    assert eq(quad_h, exp(lambda i: orc.mult_low_level(q, ct1[i], ct2[i])))
    pt = quad[:, 2].contiguous()
    ext = eng.ext_prod(mext, pt, dk)
    ext_h = eng.to_host(ext)
    assert eq(ext_h, exp(lambda i: orc.ext_prod(mext, quad_h[i, 2], key)))
    assert eq(eng.to_host(eng.ckks_rescale(mext, ext)), exp(lambda i: orc.ckks_rescale(mext, ext_h[i])))
    assert eq(eng.to_host(eng.ckks_rescale(q, d1)), exp(lambda i: orc.ckks_rescale(q, ct1[i])))
    for t in (65537, 2, 1):
        assert eq(eng.to_host(eng.bgv_mod_switch(q, t, d2)), exp(lambda i: orc.bgv_mod_drop(q, t, ct2[i])))
    assert eq(eng.to_host(eng.ckks_relinearize(mext, quad, dk)), exp(lambda i: orc.ckks_relinearize(mext, quad_h[i], key)))
    assert eq(eng.to_host(eng.bgv_relinearize(mext, quad, dk)), exp(lambda i: orc.bgv_relinearize(mext, quad_h[i], key)))
    for step in (1, 3):
        assert eq(eng.to_host(eng.ckks_rotate(mext, d1, dk, step)), exp(lambda i: orc.ckks_rotate(mext, ct1[i], key, step)))
    assert eq(eng.to_host(eng.ckks_conjugate(mext, d2, dk)), exp(lambda i: orc.ckks_conjugate(mext, ct2[i], key)))
    assert eq(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), exp(lambda i: orc.ckks_mult(mext, ct1[i], ct2[i], key)))
    assert eq(eng.to_host(eng.bgv_mult(mext, P.C5_T, d1, d2, dk)), exp(lambda i: orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key)))


def test_c3_shape_ckks_mult(eng, orc):
    """BASELINE config 3 shape (N=32768, L=10), two ciphertext pairs, raw words vs the oracle."""
    logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
    n, L, B = 1 << logn, 10, 2
    rng = SplitMix(3)
    ct1 = rng.poly((B, 2, L, n), mext[:L])
    ct2 = rng.poly((B, 2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    out = eng.to_host(eng.ckks_mult(mext, eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)))
    for i in range(B):
        assert eq(out[i], orc.ckks_mult(mext, ct1[i], ct2[i], key))


def test_error_behaviour(eng):
    from hehub_amd.engine import InvalidArgument

    x = eng.empty((1, 1, 8))
    with pytest.raises(InvalidArgument, match="Unable to drop the only one prime"):
        eng.ckks_rescale([P.P40[0]], eng.empty((1, 2, 1, 8)))
    with pytest.raises(InvalidArgument, match="2N doesn't divide"):
        eng.ntt_([12289], eng.empty((1, 1, 1 << 15)))
    with pytest.raises(InvalidArgument):
        eng.poly_scalar_mul([65537], x, [1, 2])


# ---- either side of the path ----------------------------------------------
@pytest.mark.parametrize("logn,moduli,B", [(3, P.P40[:2], 3), (8, [P.P50[1]] + P.P40[:2], 2), (12, P.P40[:4], 3),
                                           (13, P.C5_Q, 2), (15, P.C3_Q, 1)])
def test_dev_encrypt_decrypt_cores(eng, orc, logn, moduli, B):
    import torch

    rng = SplitMix(700 + logn)
    cases = [P.edge_case(rng, logn, moduli) for _ in range(B)]
    sk = cases[0][3]                                             # one secret key for the batch
    noise = np.stack([c[0] for c in cases]); c1 = np.stack([c[1] for c in cases]); pt = np.stack([c[2] for c in cases])
    d_noise = torch.from_numpy(noise).to("cuda:0")
    ct = eng.rlwe_encrypt_core(moduli, d_noise, eng.to_device(c1), eng.to_device(pt), eng.to_device(sk))
    exp = np.stack([orc.rlwe_encrypt_core(moduli, noise[i], c1[i], pt[i], sk) for i in range(B)])
    assert np.array_equal(eng.to_host(ct), exp)
    back = eng.rlwe_decrypt_core(moduli, ct, eng.to_device(sk))
    assert np.array_equal(eng.to_host(back), np.stack([orc.rlwe_decrypt_core(moduli, exp[i], sk) for i in range(B)]))


@pytest.mark.parametrize("n", [8, 4096, 32768])
def test_dev_rns_base_transforms(eng, orc, n):
    rng = SplitMix(800 + n)
    B = 3
    for old, new in ((65537, P.P40[:3]), (P.P50[0], P.P40[:3]), (P.P40[0], [65537, P.P50[1]])):
        x = np.stack([rng.words(n, 2 * old) for _ in range(B)])
        got = eng.to_host(eng.rns_base_from_single(old, new, eng.to_device(x)))
        assert np.array_equal(got, np.stack([orc.rns_base_from_single(old, new, x[i]) for i in range(B)]))
    for old, new in ((P.P40[:3], 65537), ([P.P50[0], P.P40[1]], 65537), (P.P40[:2], P.P50[2])):
        x = np.stack([P.small_rns_poly(rng, n, old) for _ in range(B)])
        x[1] = rng.poly((len(old), n), old)                      # polynomial 1 is not small: flagged, CRT stays on the host
        out, flags = eng.rns_base_to_single_small(old, new, eng.to_device(x))
        flags = flags.cpu().numpy()
        assert flags[0] == 0 and flags[1] != 0 and flags[2] == 0
        assert not orc.rns_base_to_single_small(old, new, x[1])[0]
        for i in (0, 2):
            ok, exp = orc.rns_base_to_single_small(old, new, x[i])
            assert ok and np.array_equal(eng.to_host(out)[i], exp)


def test_abi_rejects_bad_arguments(eng):
    """Status codes of the C ABI for misuse: nothing may crash or silently compute on nonsense."""
    import ctypes as C

    from hehub_amd import capi
    from hehub_amd.engine import HpError, InvalidArgument, _u64arr

    lib, h = eng.lib, eng.h
    q40 = P.P40
    buf = eng.empty((2, 2, 3, 1 << 11))
    ptr = eng._ptr(buf)
    # ring degree out of range, zero / too many limbs
    assert lib.hp_dev_ntt(h, 17, 1, _u64arr(q40[:1]), 1, ptr) == capi.HP_EUNSUPPORTED   # N = 131072: beyond the reference's 16-bit bit reversal (permutation.h:41-55)
    assert lib.hp_dev_ntt(h, 0, 1, _u64arr(q40[:1]), 1, ptr) == capi.HP_EUNSUPPORTED
    assert lib.hp_dev_ckks_mult_relin_rescale(h, 11, 1, _u64arr(q40[:2]), 1, ptr, ptr, ptr, ptr) == capi.HP_EINVAL   # only one prime
    assert lib.hp_dev_ext_prod_montgomery(h, 11, 40, _u64arr(q40[:1] * 41), 1, ptr, ptr, ptr) == capi.HP_EINVAL
    # a modulus the transform cannot use: 60 bits, or 2N does not divide q - 1
    with pytest.raises(InvalidArgument, match="59"):
        eng.ntt_([(1 << 60) - 93], eng.empty((1, 1, 8)))
    with pytest.raises(InvalidArgument, match="2N doesn't divide"):
        eng.ckks_rescale([12289, 40961], eng.empty((1, 2, 2, 1 << 14)))
    # BGV with plain modulus 0, limb ranges outside the ciphertext
    assert lib.hp_dev_bgv_mod_switch(h, 11, 3, _u64arr(q40[:3]), 0, 1, ptr, ptr) == capi.HP_EINVAL
    assert lib.hp_dev_mult_low_level_range(h, 11, 3, _u64arr(q40[:3]), 1, 2, 5, ptr, ptr, ptr) == capi.HP_EINVAL
    assert lib.hp_dev_drop_apply_range(h, 11, 3, _u64arr(q40[:3]), 0, 2, 1, 3, ptr, ptr, None, 0, 0, 0, ptr) == capi.HP_EINVAL
    msg = lib.hp_last_error(h)
    assert msg and b"range" in msg
    # NULL operands are rejected before anything is launched
    assert lib.hp_dev_ntt(h, 11, 3, _u64arr(q40[:3]), 1, None) == capi.HP_EINVAL
    assert lib.hp_dev_ckks_mult_relin_rescale(h, 11, 3, _u64arr(q40[:3] + [P.P50[0]]), 1, ptr, None, ptr, ptr) == capi.HP_EINVAL
    mis = C.c_void_p(buf.data_ptr() + 8)
    assert lib.hp_dev_ntt(h, 11, 3, _u64arr(q40[:3]), 1, mis) == capi.HP_EINVAL          # 8-byte aligned only
    # empty batches are no-ops, not errors
    assert lib.hp_dev_ntt(h, 11, 3, _u64arr(q40[:3]), 0, ptr) == capi.HP_OK
    assert lib.hp_dev_ckks_rescale(h, 11, 3, _u64arr(q40[:3]), 0, ptr, ptr) == capi.HP_OK
    # the engine is still healthy afterwards
    x = eng.to_device(np.arange(8, dtype=np.uint64).reshape(1, 1, 8))
    eng.ntt_([65537], x); eng.intt_([65537], x, strict=True)
    assert np.array_equal(eng.to_host(x).ravel(), np.arange(8, dtype=np.uint64))


@pytest.mark.parametrize("n", [8, 4096, 32768])
def test_dev_rns_base_to_single_complete(eng, orc, n):
    """rns_transform.cpp:106-127, one new modulus: small and CRT branches decided per polynomial on the device."""
    rng = SplitMix(850 + n)
    for old, new in ((P.P40[:3], 65537), ([P.P50[0], P.P40[1]], 65537), (P.P40[:5], 257), (P.P40[:2], P.P50[2]), (P.C3_Q, 786433)):
        x = np.stack([P.small_rns_poly(rng, n, old), rng.poly((len(old), n), old), rng.poly((len(old), n), old)])
        x[1, 0, 0] += np.uint64(old[0])                             # a lazy word
        x[2, :, 0] = [(q - new % q) % q for q in old]               # x = -new: the reference returns new itself, not 0
        got = eng.to_host(eng.rns_base_to_single(old, new, eng.to_device(x)))
        exp = np.stack([orc.rns_base_to_single(old, new, x[i]) for i in range(3)])
        assert np.array_equal(got, exp) and got[2, 0] == new


def test_workspace_grows_and_can_be_released(eng, orc):
    mext = P.P40[:3] + [P.P50[0]]
    rng = SplitMix(31)
    n, L = 1 << 11, 3
    ct1 = np.stack([rng.poly((2, L, n), mext[:L])]); ct2 = np.stack([rng.poly((2, L, n), mext[:L])])
    key = rng.poly((L, 2, L + 1, n), mext)
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    exp = orc.ckks_mult(mext, ct1[0], ct2[0], key)
    assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk))[0], exp)
    assert eng.lib.hp_ctx_workspace_bytes(eng.h) > 0
    eng._chk(eng.lib.hp_ctx_release_workspace(eng.h))
    assert eng.lib.hp_ctx_workspace_bytes(eng.h) == 0
    assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk))[0], exp)      # it simply grows again


def _ntt_primes(count, logn, bits=40):
    """the first `count` primes q = 1 (mod 2N) below 2^bits (deterministic Miller-Rabin for 64-bit)"""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2; r += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    step, q, out = 2 << logn, ((1 << bits) // (2 << logn)) * (2 << logn) + 1, []
    while len(out) < count:
        q -= step
        if is_prime(q):
            out.append(q)
    return out


@pytest.mark.parametrize("logn", [6, 11])
def test_maximum_number_of_moduli(eng, orc, logn):
    """HP_MAX_LIMBS = 32: a ciphertext of 31 moduli plus the special prime through the whole pipeline (generic and
    tiled kernels, the latter with the fused drop-last-prime constants at their full width)."""
    L = 31
    mext = _ntt_primes(L + 1, logn)
    n = 1 << logn
    rng = SplitMix(4242)
    ct1 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(2)]); ct2 = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(2)])
    key = rng.poly((L, 2, L + 1, n), mext)
    d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
    assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(2)]))
    assert np.array_equal(eng.to_host(eng.bgv_mult(mext, 65537, d1, d2, dk)),
                          np.stack([orc.bgv_mult(mext, 65537, ct1[i], ct2[i], key) for i in range(2)]))
    assert np.array_equal(eng.to_host(eng.ckks_rotate(mext, d1, dk, 5)), np.stack([orc.ckks_rotate(mext, ct1[i], key, 5) for i in range(2)]))
    from hehub_amd.engine import InvalidArgument

    with pytest.raises(InvalidArgument):      # one more modulus does not fit
        eng.ckks_rotate(mext + _ntt_primes(L + 2, logn)[-1:], eng.empty((1, 2, L + 1, n)), eng.empty((L + 1, 2, L + 2, n)), 1)


@pytest.mark.parametrize("logn,L,B", [(4, 3, 3), (11, 4, 5), (12, 10, 2), (11, 31, 2), (11, 1, 5), (12, 2, 7), (3, 1, 2)])
def test_ext_prod_with_arbitrary_words(eng, orc, logn, L, B):
    """rgsw.cpp:121-153 accumulates in wrapping u128 and never looks at the size of a word: inputs and key words that use
    all 64 bits (no valid ciphertext has them) must still give the reference's words -- this is what drives every carry
    path of the column accumulators (hp_device.h: hp_mac2), including the one 50-bit moduli never reach."""
    mext = _ntt_primes(L + 1, logn) if L + 1 > len(P.P40) else P.P40[:L] + [P.P50[0]]
    n = 1 << logn
    rng = SplitMix(4242 + logn + L)
    pt = rng.words(B * L * n).reshape(B, L, n)
    key = rng.words(L * 2 * (L + 1) * n).reshape(L, 2, L + 1, n)
    key[:, :, :, ::3] = np.uint64(0xFFFFFFFFFFFFFFFF)      # runs of maximal words: the carries pile up
    pt[:, :, 1::3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    got = eng.to_host(eng.ext_prod(mext, eng.to_device(pt), eng.to_device(key)))
    for i in range(B):
        assert np.array_equal(got[i], orc.ext_prod(mext, pt[i], key)), i


def test_pipeline_is_capturable_in_a_hip_graph(eng, orc):
    """After one warm-up call (tables, constants and workspace exist) an entry point only enqueues kernels on the
    context's stream: it can be captured into a HIP graph and replayed on new inputs (include/hehub_amd.h, hp_ctx_set_stream)."""
    import torch

    logn, L = 11, 3
    mext = P.P40[:L] + [P.P50[0]]
    n, B = 1 << logn, 3
    rng = SplitMix(777)
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
            eng.ckks_mult(mext, d1, d2, dk, out=out)            # warm-up on the capture stream
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                eng.ckks_mult(mext, d1, d2, dk, out=out)
        torch.cuda.current_stream().wait_stream(side)
        exp = lambda a, b: np.stack([orc.ckks_mult(mext, a[i], b[i], key) for i in range(B)])
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(eng.to_host(out), exp(ct1, ct2))
        d1.copy_(eng.to_device(ct2)); d2.copy_(eng.to_device(ct1))   # new inputs in the captured buffers
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(eng.to_host(out), exp(ct2, ct1))
    finally:
        eng.use_stream(torch.cuda.current_stream())


def test_stream_switches_keep_device_order(eng, orc):
    """all calls of a context share one scratch workspace, so switching the context to another stream must
    not let new work overtake what is queued on the previous one.  Calls alternate between two side streams with no
    host synchronisation in between; every result must equal the oracle's."""
    import torch

    logn, L = 13, 4
    mext = P.P40[:L] + [P.P50[0]]
    n, B, rounds = 1 << logn, 24, 6
    rng = SplitMix(4242)
    base1 = rng.poly((3, 2, L, n), mext[:L]); base2 = rng.poly((3, 2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    idx = torch.arange(B, device="cuda:0") % 3
    d1 = eng.to_device(base1).index_select(0, idx).contiguous(); d2 = eng.to_device(base2).index_select(0, idx).contiguous()
    dk = eng.to_device(key)
    exp = [np.stack([orc.ckks_mult(mext, base1[c], base2[c], key) for c in range(3)]),
           np.stack([orc.ckks_mult(mext, base2[c], base2[c], key) for c in range(3)])]
    outs = [eng.empty((B, 2, L - 1, n)) for _ in range(rounds)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    try:
        for r in range(rounds):
            eng.use_stream(streams[r & 1])
            eng.ckks_mult(mext, d1 if r % 2 == 0 else d2, d2, dk, out=outs[r])
        torch.cuda.synchronize()
    finally:
        eng.use_stream(torch.cuda.current_stream())
    for r in range(rounds):
        got = eng.to_host(outs[r])
        for i in range(B):
            assert np.array_equal(got[i], exp[r % 2][i % 3]), (r, i)


def test_workspace_generation_and_per_thread_errors(eng):
    """the workspace generation counter moves when the scratch block is replaced (captured graphs go stale); an error
    message belongs to the thread whose call failed"""
    import threading

    from hehub_amd.engine import InvalidArgument

    eng.release_workspace()
    g0 = eng.lib.hp_ctx_workspace_generation(eng.h)
    eng.ckks_rescale(P.P40[:3], eng.empty((1, 2, 3, 2048)))
    g1 = eng.lib.hp_ctx_workspace_generation(eng.h)
    eng.ckks_rescale(P.P40[:3], eng.empty((1, 2, 3, 2048)))          # same shape: no reallocation
    assert g1 > g0 and eng.lib.hp_ctx_workspace_generation(eng.h) == g1
    eng.ckks_rescale(P.P40[:3], eng.empty((4, 2, 3, 2048)))          # larger: replaced
    assert eng.lib.hp_ctx_workspace_generation(eng.h) > g1
    msgs = {}

    def worker(name, fn):
        try:
            fn()
        except InvalidArgument as e:
            msgs[name] = e.msg

    t1 = threading.Thread(target=worker, args=("drop", lambda: eng.ckks_rescale([P.P40[0]], eng.empty((1, 2, 1, 8)))))
    t2 = threading.Thread(target=worker, args=("ntt", lambda: eng.ntt_([12289], eng.empty((1, 1, 1 << 15)))))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert "Unable to drop the only one prime" in msgs["drop"] and "2N doesn't divide" in msgs["ntt"]


def test_digit_workspace_packing_is_invisible(eng, orc):
    """At N = 32768 the digit rows of output moduli whose folded words are provably below 2^48 cross HBM as 6 bytes per word
    (HP_PACK48, hp_device.h) -- an internal workspace format.  Mixed chain: 40-bit list primes (packed), a 50-bit prime and a
    prime far from a power of two, 1.37 * 2^40, whose fold wraps exactly as the reference's does (both must stay plain rows);
    the results equal the oracle's and those of a context created with HP_NO_PACK48."""
    import os

    from hehub_amd.engine import Engine

    far = 1506330935297                      # = 1 (mod 2^16), prime, 37 % above 2^40
    logn, B = 15, 3
    for mext in ([P.P40[0], far, P.P50[1], P.P40[1], P.P50[0]], [P.P40[2], P.P40[3], far]):
        n, L = 1 << logn, len(mext) - 1
        rng = SplitMix(4848 + L)
        pt = rng.poly((B, L, n), mext[:L])
        ct1 = rng.poly((B, 2, L, n), mext[:L]); ct2 = rng.poly((B, 2, L, n), mext[:L])
        key = rng.poly((L, 2, L + 1, n), mext)
        exp_ext = np.stack([orc.ext_prod(mext, pt[i], key) for i in range(B)])
        exp_mul = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)])
        os.environ["HP_NO_PACK48"] = "1"
        try:
            plain = Engine(0)
        finally:
            del os.environ["HP_NO_PACK48"]
        try:
            for e in (eng, plain):
                dk = e.to_device(key)
                assert np.array_equal(e.to_host(e.ext_prod(mext, e.to_device(pt), dk)), exp_ext)
                assert np.array_equal(e.to_host(e.ckks_mult(mext, e.to_device(ct1), e.to_device(ct2), dk)), exp_mul)
        finally:
            plain.close()


def test_dev_copy(eng):
    """hp_dev_copy: the deep copy of device words (allocator.h:113-118 for HBM-resident limbs); odd counts, overlap, NULL"""
    import torch

    from hehub_amd.engine import InvalidArgument

    for words in (1, 2, 3, 1023, 1024, 2048 * 4 + 2, (1 << 20) + 7):
        src = torch.randint(-2**62, 2**62, (words + 2,), dtype=torch.int64, device="cuda:0")
        dst = torch.full((words + 4,), 7, dtype=torch.int64, device="cuda:0")
        eng._chk(eng.lib.hp_dev_copy(eng.h, words, eng._ptr(src), eng._ptr(dst[2:])))     # (dst + 2 words: 16-byte aligned)
        assert torch.equal(dst[2:2 + words], src[:words]) and (dst[:2] == 7).all() and (dst[2 + words:] == 7).all()
    buf = torch.zeros(4096, dtype=torch.int64, device="cuda:0")
    with pytest.raises(InvalidArgument, match="overlap"):
        eng._chk(eng.lib.hp_dev_copy(eng.h, 2048, eng._ptr(buf), eng._ptr(buf[1024:])))
    with pytest.raises(InvalidArgument, match="aligned"):
        eng._chk(eng.lib.hp_dev_copy(eng.h, 8, eng._ptr(buf), eng._ptr(buf[1:])))
    eng._chk(eng.lib.hp_dev_copy(eng.h, 0, eng._ptr(buf), eng._ptr(buf)))
    # the Python wrapper refuses an `out` the kernel would overrun or misread: short, strided, 4-byte words, host memory
    src = torch.arange(64, dtype=torch.int64, device="cuda:0")
    for bad in (torch.zeros(32, dtype=torch.int64, device="cuda:0"), torch.zeros(128, dtype=torch.int64, device="cuda:0")[::2],
                torch.zeros(64, dtype=torch.int32, device="cuda:0"), torch.zeros(64, dtype=torch.int64)):
        with pytest.raises(InvalidArgument, match="copy:"):
            eng.copy(src, out=bad)
    assert torch.equal(eng.copy(src), src)


def test_host_rows_over_pcie(eng):
    """hp_host_register + hp_dev_store_host_rows / hp_dev_load_host_rows: a polynomial whose limbs are separate blocks of
    caller-owned host memory (rns.h:15-156: one SmartArray per limb, allocator.h:105-223) crosses PCIe by one kernel"""
    import ctypes as C

    import torch

    from hehub_amd import capi
    from hehub_amd.engine import HpError, InvalidArgument

    lib, h = eng.lib, eng.h
    rows, n = 7, 32768
    blocks = [np.zeros(n + 512, dtype=U) for _ in range(rows)]          # separate allocations, 256 KiB + slack each
    views = []
    for b in blocks:
        off = (-b.ctypes.data // 8) % 2                                  # 16-byte aligned start
        views.append(b[off:off + n])
        assert lib.hp_host_register(h, C.c_void_p(b.ctypes.data), b.nbytes) == capi.HP_OK
    ptrs = (C.c_void_p * rows)(*[v.ctypes.data for v in views])
    src = torch.randint(-2**62, 2**62, (rows, n), dtype=torch.int64, device="cuda:0")
    eng._chk(lib.hp_dev_store_host_rows(h, rows, n, eng._ptr(src), ptrs))
    eng.sync()
    host = src.cpu().numpy().view(U)
    for r in range(rows):
        assert np.array_equal(views[r], host[r])
        views[r][:] = views[r] ^ U(0x5555)                                # the host changes its words ...
    back = torch.zeros_like(src)
    eng._chk(lib.hp_dev_load_host_rows(h, rows, n, eng._ptr(back), ptrs))
    eng.sync()
    assert np.array_equal(back.cpu().numpy().view(U), host ^ U(0x5555))  # ... and the device reads them
    # a row that is plain pageable memory is refused, not read through a wild pointer
    loose = np.zeros(n, dtype=U)
    bad = (C.c_void_p * 1)(loose.ctypes.data)
    with pytest.raises(InvalidArgument, match="not registered"):
        eng._chk(lib.hp_dev_store_host_rows(h, 1, n, eng._ptr(src), bad))
    with pytest.raises(InvalidArgument, match="even"):
        eng._chk(lib.hp_dev_store_host_rows(h, rows, n - 1, eng._ptr(src), ptrs))
    # asynchronous copies from / to a registered block
    eng._chk(lib.hp_memcpy_d2h_async(h, C.c_void_p(views[0].ctypes.data), eng._ptr(src), n * 8))
    eng.sync()
    assert np.array_equal(views[0], host[0])
    eng._chk(lib.hp_memcpy_h2d_async(h, eng._ptr(back), C.c_void_p(views[0].ctypes.data), n * 8))
    eng.sync()
    assert np.array_equal(back[0].cpu().numpy().view(U), host[0])
    for b in blocks:
        assert lib.hp_host_unregister(h, C.c_void_p(b.ctypes.data)) == capi.HP_OK
