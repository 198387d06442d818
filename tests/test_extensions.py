"""Extensions beyond the reference (include/hehub_amd.h): a key generated at a higher level used
at a lower one, and rescale by several primes.  hehub throws in both cases, so they are pinned by equivalence: the
result must equal, word for word, what the reference-parity entry points (and the oracle) give for the extracted
sub-key / for successive single drops."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu


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
