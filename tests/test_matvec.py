"""hehub's circuit-level caller of the key switch: the diagonal loop of matrix_vector_mul_short (src/circuits/linear_algebra.h:104-136)
-- per diagonal a plaintext product, an accumulate, and (matrix narrower than the slot count) TWO rotations of the input vector, each
under its own key of the rotation key set.  examples/diag_matvec.cpp makes those calls in that order on synthetic words; it must print
hehub's own digest (tests/golden/matvec.json, generated from hehub itself by tests/golden/make_matvec.py; where the prebuilt
oracle/_ref/ref_matvec_cpu exists, also that program run on the spot) however the layer runs it: eagerly over lanes, recorded (deferred
mode: the rotations under different keys run as ONE hp_dev_ckks_rotate_many sequence), and written with the batched form
amd::rotate(cts, keys, steps)."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_matvec import CASES, REF, run  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "matvec.json")) as f:
    GOLDEN = json.load(f)["digests"]
REF_AMD = os.path.join(ROOT, "oracle", "_ref", "ref_matvec_amd")


def binary():
    from hehub_amd.build import build_example

    return build_example("diag_matvec")


def key(case):
    return " ".join(str(a) for a in case)


def test_every_case_has_hehubs_digest():
    assert set(GOLDEN) == {key(c) for c in CASES}
    if os.path.exists(REF):      # hehub itself, here and now
        assert run(REF, CASES[0])[0]["loop"] == GOLDEN[key(CASES[0])]
        assert run(REF, CASES[7])[0]["loop"] == GOLDEN[key(CASES[7])]


def test_example_builds():
    assert os.path.exists(binary())


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(a) for a in c))
def test_matvec_prints_hehubs_digest_in_every_mode(case):
    want = GOLDEN[key(case)]
    envs = ({}, {"HEHUB_AMD_LANES": "1"}, {"HEHUB_AMD_KEY_CACHE": "2"}) if case[0] <= 13 else ({},)   # (the big shapes once: their setup is host time)
    for env in envs:
        got, _, text = run(binary(), case, env, reps=2)
        modes = ("eager", "deferred") + (("batched-form",) if case[3] == "short" else ())
        for m in modes:
            assert got[m] == want, (m, env, case, got, want, text)
        if case[3] == "short" and case[2] > 2:
            assert "many_key_groups 0" not in text, text       # the rotations under different keys really ran as one sequence
    if os.path.exists(REF_AMD) and case[0] <= 13:   # hehub's own objects over the binding: every call crosses PCIe, small rings only
        assert run(REF_AMD, case)[0]["loop"] == want, ("binding", case)


@pytest.mark.gpu
def test_matvec_modes_agree_at_parity_level_a():
    """canonical residues have one representative: eager, deferred (rotate_many) and the batched form agree with each other"""
    for case in (CASES[1], CASES[2], CASES[8]):
        got, _, text = run(binary(), case, {"HP_PARITY_LEVEL": "A"}, reps=2)
        assert len(set(got.values())) == 1, (case, got)
        assert "parity level A" in text
        # (the last call of the loop is a rescale: its Harvey product by a constant leaves a word >= q with probability ~ q / 2^64, so
        # hehub's lazy words of this program are nearly always canonical already -- the digests of the two levels usually coincide)


@pytest.mark.gpu
def test_matvec_rotations_under_different_keys_run_batched():
    """width 16: 30 rotations of one vector under 30 keys.  Recorded, they run as one launch sequence with a key per ciphertext
    (hp_dev_ckks_rotate_many_rows), the plaintext transforms as one batch; every mode prints hehub's digest, and the layer's own counters
    say the rotations ran grouped across keys.  (The times -- 0.7 against 2.7 ms per product vector at N = 8192, 2.1 against 2.4 at the
    C3 shape, hehub on the host CPU 1262 ms -- are held in tests/test_perf.py.)"""
    for case in ((13, 6, 16, "short"), CASES[6]):
        got, ms, text = run(binary(), case, reps=2)
        assert got["eager"] == got["deferred"] == got["batched-form"], text
        m = re.search(r"many_key_groups (\d+)", text)
        assert m and int(m.group(1)) >= 1, text
    assert got["eager"] == GOLDEN[key(CASES[6])], text
