"""Wire / on-disk format: include/hehub_amd.h hp_wire_*, hehub_amd/csrc/hp_wire.cpp."""
import os
import struct
import sys

import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
FIXTURE = os.path.join(HERE, "golden", "ckks_mult_n8.hehubamd")


def test_round_trip_every_kind():
    from hehub_amd import wire

    rng = SplitMix(42)
    for kind, polys, logn, moduli in ((wire.POLY, 1, 5, P.P40[:2]), (wire.CT, 2, 11, P.P40[:3]), (wire.QUAD_CT, 3, 4, P.P50[:1]),
                                      (wire.KSK, 6, 6, P.P40[:3] + [P.P50[0]])):
        words = rng.poly((polys, len(moduli), 1 << logn), moduli)
        blob = wire.pack(kind, moduli, words, rep_form=1, scalar=65537 if kind != wire.CT else 2.0 ** 40)
        d, m, w = wire.unpack(blob)
        assert (d.kind, d.polys, d.limbs, d.log_dimension, d.rep_form) == (kind, polys, len(moduli), logn, 1)
        assert m == list(moduli) and np.array_equal(w, words)
        assert len(blob) == 40 + 8 * len(moduli) + 8 * words.size + 8
        if kind == wire.CT:
            assert struct.unpack("<d", struct.pack("<Q", d.scheme_scalar))[0] == 2.0 ** 40


def test_corruption_is_detected():
    from hehub_amd import wire

    rng = SplitMix(43)
    blob = bytearray(wire.pack(wire.CT, P.P40[:2], rng.poly((2, 2, 16), P.P40[:2])))
    for pos in (0, 9, 13, 21, 45, 100, len(blob) - 1):
        bad = bytearray(blob)
        bad[pos] ^= 0x10
        with pytest.raises(ValueError):
            wire.unpack(bytes(bad))
    with pytest.raises(ValueError):
        wire.unpack(bytes(blob[:-8]))
    with pytest.raises(ValueError):
        wire.pack(wire.CT, P.P40[:2], rng.poly((3, 2, 16), P.P40[:2]))     # a ciphertext has exactly two polynomials
    wire.unpack(bytes(blob))


def test_committed_fixture_carries_the_reference_result(orc):
    """The fixture was written by tests/golden/make_golden.py from the compiled reference: same bytes when re-packed,
    same words as the oracle computes."""
    from cases import wire_fixture_case
    from hehub_amd import wire

    blob = open(FIXTURE, "rb").read()
    d, moduli, words = wire.unpack(blob)
    mext, ct1, ct2, key = wire_fixture_case()
    assert moduli == mext[:2] and d.kind == wire.CT and d.log_dimension == 3
    assert np.array_equal(words, orc.ckks_mult(mext, ct1, ct2, key))
    assert wire.pack(wire.CT, moduli, words, rep_form=1, scalar=2.0 ** 30) == blob


@pytest.mark.gpu
def test_device_load_and_store(orc):
    from hehub_amd import wire
    from hehub_amd.engine import Engine

    eng = Engine(0)
    rng = SplitMix(44)
    logn, mext = 11, P.P40[:3] + [P.P50[0]]
    n, L = 1 << logn, 3
    ct = rng.poly((2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    d, m, d_ct = wire.load_to_device(eng, wire.pack(wire.CT, mext[:L], ct))
    _, _, d_key = wire.load_to_device(eng, wire.pack(wire.KSK, mext, key.reshape(2 * L, L + 1, n)))
    out = eng.ckks_rotate(mext, d_ct.view(1, 2, L, n), d_key.view(L, 2, L + 1, n), 1)
    blob = wire.store_from_device(eng, wire.CT, mext[:L], out[0])
    _, m2, words = wire.unpack(blob)
    assert m2 == mext[:L] and np.array_equal(words, orc.ckks_rotate(mext, ct, key, 1))
    eng.close()
