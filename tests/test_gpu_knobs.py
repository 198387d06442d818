"""The engine's tuning knobs (environment variables read once at hp_ctx_create, hp_ctx.cpp) change schedules and workspace
formats, never results: every knob at its extremes and at out-of-range values (which are clamped) must give the oracle's words
("none is validated for range in tests")."""
import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

pytestmark = pytest.mark.gpu

KNOBS = [
    {},
    {"HP_SPREAD_GROUP": "0"}, {"HP_SPREAD_GROUP": "1"}, {"HP_SPREAD_GROUP": "3"}, {"HP_SPREAD_GROUP": "1000"}, {"HP_SPREAD_GROUP": "-5"},
    {"HP_DROP_GROUP": "0"}, {"HP_DROP_GROUP": "1"}, {"HP_DROP_GROUP": "7"}, {"HP_DROP_GROUP": "99999"},
    {"HP_NO_PACK48": "1"}, {"HP_PACK48_MIN_LOGN": "0"}, {"HP_PACK48_MIN_LOGN": "15"}, {"HP_PACK48_MIN_LOGN": "99"},
    {"HP_NO_FUSED_DROP": "1"},
    {"HP_MULT_STREAMS": "2"}, {"HP_MULT_STREAMS": "2", "HP_MULT_CHUNK": "3"}, {"HP_MULT_CHUNK": "1"}, {"HP_MULT_CHUNK": "-1"},
    {"HP_MULT_STREAMS": "7", "HP_MULT_CHUNK": "100000000000"},
    {"HP_SPLIT_MAX_ITEMS": "0"}, {"HP_SPLIT_MAX_ITEMS": "4096"}, {"HP_SPLIT_MAX_ITEMS": "-3"}, {"HP_SPLIT_MAX_ITEMS": "20"},
    {"HP_SPLIT_MAX_ITEMS": "4096", "HP_NO_FUSED_DROP": "1"},
]


@pytest.fixture(scope="module")
def cases(orc):
    out = []
    for logn, mext, B, seed in ((12, [P.P50[1]] + P.P40[:4] + [P.P50[0]], 5, 801), (15, P.C3_MODULI_EXT, 2, 802)):
        n, L = 1 << logn, len(mext) - 1
        rng = SplitMix(seed)
        ct1, ct2 = rng.poly((B, 2, L, n), mext[:L]), rng.poly((B, 2, L, n), mext[:L])
        key = rng.poly((L, 2, L + 1, n), mext)
        exp = {"ckks": np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)]),
               "bgv": np.stack([orc.bgv_mult(mext, P.C5_T, ct1[i], ct2[i], key) for i in range(B)]),
               "rot": np.stack([orc.ckks_rotate(mext, ct1[i], key, 3) for i in range(B)])}
        out.append((mext, ct1, ct2, key, exp))
    return out


@pytest.mark.parametrize("env", KNOBS, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "defaults")
def test_knob_settings_give_the_same_words(cases, monkeypatch, env):
    from hehub_amd.engine import Engine

    for k in ("HP_SPREAD_GROUP", "HP_DROP_GROUP", "HP_NO_PACK48", "HP_PACK48_MIN_LOGN", "HP_NO_FUSED_DROP", "HP_MULT_STREAMS",
              "HP_MULT_CHUNK", "HP_SPLIT_MAX_ITEMS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = Engine(0)      # the knobs are read here
    try:
        for mext, ct1, ct2, key, exp in cases:
            d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
            assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), exp["ckks"]), env
            assert np.array_equal(eng.to_host(eng.bgv_mult(mext, P.C5_T, d1, d2, dk)), exp["bgv"]), env
            assert np.array_equal(eng.to_host(eng.ckks_rotate(mext, d1, dk, 3)), exp["rot"]), env
    finally:
        eng.close()


KNOBS_A = [{}, {"HP_SPREAD_GROUP": "0"}, {"HP_SPREAD_GROUP": "3"}, {"HP_DROP_GROUP": "0"}, {"HP_DROP_GROUP": "7"}, {"HP_NO_PACK48": "1"},
           {"HP_PACK48_MIN_LOGN": "15"}, {"HP_MULT_STREAMS": "2", "HP_MULT_CHUNK": "3"}, {"HP_MULT_CHUNK": "1"}, {"HP_NO_FUSED_DROP": "1"}, {"HP_NO_DOUBLE_DROP": "1"}, {"HP_NO_DOUBLE_DROP": "1", "HP_DROP_GROUP": "0"}, {"HP_NO_PACK40": "1"}]


@pytest.mark.parametrize("env", KNOBS_A, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "defaults")
def test_knob_settings_at_parity_level_a(cases, monkeypatch, env):
    """the same with HP_PARITY_LEVEL=A in the environment: every schedule knob gives the canonical residues of the oracle's words;
    HP_NO_FUSED_DROP (a debug path through the unfused level-B drop kernels) keeps the whole call at level B"""
    from hehub_amd.engine import Engine

    for k in ("HP_SPREAD_GROUP", "HP_DROP_GROUP", "HP_NO_PACK48", "HP_PACK48_MIN_LOGN", "HP_NO_FUSED_DROP", "HP_MULT_STREAMS",
              "HP_MULT_CHUNK", "HP_NO_DOUBLE_DROP", "HP_NO_PACK40"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("HP_PARITY_LEVEL", "A")
    eng = Engine(0)
    try:
        assert eng.parity_level() == "A"
        for mext, ct1, ct2, key, exp in cases:
            fin = (lambda a: a) if "HP_NO_FUSED_DROP" in env else (lambda a: a % np.array(mext[:a.shape[-2]], dtype=np.uint64)[:, None])
            d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
            assert np.array_equal(eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), fin(exp["ckks"])), env
            assert np.array_equal(eng.to_host(eng.bgv_mult(mext, P.C5_T, d1, d2, dk)), fin(exp["bgv"])), env
            assert np.array_equal(eng.to_host(eng.ckks_rotate(mext, d1, dk, 3)), fin(exp["rot"])), env
    finally:
        eng.close()


@pytest.mark.parametrize("logn", [12, 13, 14, 15, 16])
def test_split_limb_transforms_are_the_reference_words(orc, monkeypatch, logn):
    """hp_ntt_split.hip (round 5): a launch of few limbs cuts every limb into N / 2048 tiles and runs the
    reference's stages in two launches of small workgroups -- the same butterflies in another order (ntt.cpp:155-176, :178-223), so
    the raw words must be hehub's: forward, inverse, strict inverse, against the oracle and against the tiled kernels
    (HP_SPLIT_MAX_ITEMS=0) where those exist."""
    from hehub_amd.engine import Engine

    moduli = _moduli_for(logn)
    n, L, B = 1 << logn, len(moduli), 3
    rng = SplitMix(9100 + logn)
    x = rng.poly((B, L, n), moduli)
    x[0, :, :5] = (np.array(moduli, dtype=np.uint64) - np.uint64(1))[:, None]
    fwd = np.stack([orc.poly_ntt(moduli, x[i]) for i in range(B)])
    inv = np.stack([orc.poly_intt(moduli, fwd[i]) for i in range(B)])
    got = {}
    for setting in ("4096", "0"):
        monkeypatch.setenv("HP_SPLIT_MAX_ITEMS", setting)
        eng = Engine(0)
        try:
            y = eng.ntt_(moduli, eng.to_device(x))
            f = eng.to_host(y).copy()
            z = eng.to_host(eng.intt_(moduli, y)).copy()
            s = eng.to_host(eng.intt_(moduli, eng.to_device(fwd), strict=True)).copy()
            got[setting] = (f, z, s)
        finally:
            eng.close()
        assert np.array_equal(f, fwd) and np.array_equal(z, inv), (logn, setting)
        assert np.array_equal(s, np.stack([orc.poly_reduce_strict(moduli, inv[i]) for i in range(B)])), (logn, setting)
    for a, b in zip(got["4096"], got["0"]):
        assert np.array_equal(a, b)


def _moduli_for(logn):
    if logn <= 15:
        return [P.P50[1]] + P.P40[:3]
    from test_gpu_parity import _ntt_primes
    return _ntt_primes(2, 16, 40) + _ntt_primes(1, 16, 49)
