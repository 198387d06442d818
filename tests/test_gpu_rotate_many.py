"""hp_dev_ckks_rotate_many: a batch of rotations / conjugations in which every ciphertext has its OWN key and step -- the rotations of
one vector under the keys of a rotation key set (src/circuits/linear_algebra.h:123-130).  Pinned to the oracle's single calls
(ckks/arith.cpp:75-93) word for word at parity level B, to their canonical residues at level A, and to the one-key entry point."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu
U = np.uint64


def canon(moduli, a):
    return a % np.array(moduli[:a.shape[-2]], dtype=U)[:, None]


@pytest.fixture(scope="module", params=["B", "A"])
def eng(request):
    from hehub_amd.engine import Engine

    e = Engine(0)
    e.set_parity_level(request.param)
    e.level = request.param
    yield e
    e.release_workspace()
    e.close()


# (logn, L0, L, B): generic sizes, tiled sizes, a key of a higher level, more ciphertexts than one key table holds (32)
@pytest.mark.parametrize("logn,L0,L,B", [(4, 2, 2, 3), (11, 3, 3, 5), (12, 4, 3, 7), (13, 3, 3, 35), (15, 2, 2, 4)])
def test_rotate_many_against_single_calls(eng, orc, logn, L0, L, B):
    n = 1 << logn
    q_full, p = ([P.P50[1]] + P.P40)[:L0], P.P50[0]
    mext = q_full[:L] + [p]
    rng = SplitMix(4200 + logn)
    nkeys = min(B, 6)
    keys_full = [rng.poly((L0, 2, L0 + 1, n), q_full + [p]) for _ in range(nkeys)]
    subs = [np.ascontiguousarray(k[:L][:, :, list(range(L)) + [L0], :]) for k in keys_full]
    dkeys = [eng.to_device(k) for k in keys_full]
    ct = np.stack([rng.poly((2, L, n), mext[:L]) for _ in range(B)])
    steps = [(1, 5, 1, n // 2 - 1, 3, 1)[b % 6] if b % 7 != 4 else 0 for b in range(B)]
    conj = [b % 7 == 4 for b in range(B)]
    which = [(b * 5 + 1) % nkeys for b in range(B)]
    d = eng.to_device(ct)
    got = eng.to_host(eng.ckks_rotate_many(mext, L0, d, [dkeys[w] for w in which], steps, conj))
    n_check = B if logn <= 13 else 2
    for b in list(range(n_check)) + [B - 1]:
        want = orc.ckks_conjugate(mext, ct[b], subs[which[b]]) if conj[b] else orc.ckks_rotate(mext, ct[b], subs[which[b]], steps[b])
        if eng.level == "A" and logn >= 11:
            want = canon(mext, want)
        assert np.array_equal(got[b], want), b
    # ... and the one-key entry point on each ciphertext alone writes the same words
    for b in (0, B - 1):
        one = eng.ckks_rotate_at(mext, L0, d[b:b + 1], dkeys[which[b]], steps[b]) if not conj[b] else None
        if one is not None:
            assert np.array_equal(eng.to_host(one)[0], got[b])
    # no conj array at all: plain rotations
    got2 = eng.to_host(eng.ckks_rotate_many(mext, L0, d[:2], [dkeys[which[0]], dkeys[which[1]]], [2, 2]))
    for b in range(2):
        want = orc.ckks_rotate(mext, ct[b], subs[which[b]], 2)
        assert np.array_equal(got2[b], canon(mext, want) if eng.level == "A" and logn >= 11 else want)


@pytest.mark.parametrize("logn,L,B", [(12, 3, 6), (13, 2, 40)])
def test_rotate_many_rows_one_vector_many_keys(eng, orc, logn, L, B):
    """the polynomials by address: ONE vector (two separate device tensors) rotated under B keys, plus a second vector in between"""
    n = 1 << logn
    q, p = ([P.P50[1]] + P.P40)[:L], P.P50[0]
    mext = q + [p]
    rng = SplitMix(4300 + logn)
    nkeys = min(B, 5)
    keys = [rng.poly((L, 2, L + 1, n), mext) for _ in range(nkeys)]
    dkeys = [eng.to_device(k) for k in keys]
    vec, other = rng.poly((2, L, n), q), rng.poly((2, L, n), q)
    dv = (eng.to_device(vec[0]), eng.to_device(vec[1]))
    do = eng.to_device(other)
    polys = [dv if b % 4 != 3 else (do[0], do[1]) for b in range(B)]
    steps = [b + 1 for b in range(B)]
    conj = [b % 9 == 8 for b in range(B)]
    got = eng.to_host(eng.ckks_rotate_many_rows(mext, L, polys, [dkeys[b % nkeys] for b in range(B)], steps, conj))
    for b in range(B):
        src = vec if b % 4 != 3 else other
        want = orc.ckks_conjugate(mext, src, keys[b % nkeys]) if conj[b] else orc.ckks_rotate(mext, src, keys[b % nkeys], steps[b])
        assert np.array_equal(got[b], canon(mext, want) if eng.level == "A" else want), b


def test_rotate_many_rejects_bad_arguments(eng):
    from hehub_amd.engine import InvalidArgument

    logn, L = 11, 2
    n = 1 << logn
    mext = P.P40[:L] + [P.P50[0]]
    ct, key = eng.empty((2, 2, L, n)), eng.empty((L, 2, L + 1, n))
    with pytest.raises(InvalidArgument):
        eng.ckks_rotate_many(mext, L, ct, [key, key], [1, 1 << 17])
    with pytest.raises(InvalidArgument):
        eng.ckks_rotate_many(mext, L - 1, ct, [key, key], [1, 1])
    import ctypes as C

    from hehub_amd import capi
    kp = (capi.P * 2)(key.data_ptr(), None)
    st = (C.c_size_t * 2)(1, 1)
    out = eng.empty((2, 2, L, n))
    rc = eng.lib.hp_dev_ckks_rotate_many(eng.h, logn, L, L, (capi.u64 * 3)(*mext), 2, st, None, C.c_void_p(ct.data_ptr()), kp, C.c_void_p(out.data_ptr()))
    assert rc == capi.HP_EINVAL
    kp = (capi.P * 2)(key.data_ptr(), key.data_ptr())
    pp = (capi.P * 4)(ct[0, 0].data_ptr(), ct[0, 1].data_ptr(), ct[1, 0].data_ptr(), None)
    rc = eng.lib.hp_dev_ckks_rotate_many_rows(eng.h, logn, L, L, (capi.u64 * 3)(*mext), 2, st, None, pp, kp, C.c_void_p(out.data_ptr()))
    assert rc == capi.HP_EINVAL
    rc = eng.lib.hp_dev_ckks_rotate_many_rows(eng.h, logn, L, L, (capi.u64 * 3)(*mext), 2, st, None, None, kp, C.c_void_p(out.data_ptr()))
    assert rc == capi.HP_EINVAL


@pytest.mark.parametrize("logn,L,B", [(11, 3, 3), (12, 4, 70), (13, 2, 5)])
def test_fused_mult_with_operands_by_address(eng, orc, logn, L, B):
    """hp_dev_ckks_mult_relin_rescale_rows / hp_dev_bgv_mult_relin_modswitch_rows: the operand polynomials anywhere on the device (more
    pairs than one table of 64 holds; some polynomials shared between pairs) -- the oracle's ckks::mult + rescale_inplace, word for word"""
    n = 1 << logn
    q, p = ([P.P50[1]] + P.P40)[:L], P.P50[0]
    mext = q + [p]
    rng = SplitMix(4400 + logn)
    key = rng.poly((L, 2, L + 1, n), mext)
    dk = eng.to_device(key)
    nd = min(B, 6)
    cts = [rng.poly((2, L, n), q) for _ in range(nd + 1)]
    dev = [(eng.to_device(c[0]), eng.to_device(c[1])) for c in cts]
    pairs = [(dev[b % nd][0], dev[b % nd][1], dev[(b * 3 + 1) % (nd + 1)][0], dev[(b * 3 + 1) % (nd + 1)][1]) for b in range(B)]
    got = eng.to_host(eng.ckks_mult_rows(mext, L, pairs, dk))
    t = 65537
    gotb = eng.to_host(eng.bgv_mult_rows(mext, t, pairs, dk))
    for b in sorted(set(list(range(min(B, 8))) + [B - 1])):
        a, c = cts[b % nd], cts[(b * 3 + 1) % (nd + 1)]
        want, wantb = orc.ckks_mult(mext, a, c, key), orc.bgv_mult(mext, t, a, c, key)
        lvl_a = eng.level == "A"
        assert np.array_equal(got[b], canon(mext, want) if lvl_a else want), b
        assert np.array_equal(gotb[b], canon(mext, wantb) if lvl_a else wantb), b
