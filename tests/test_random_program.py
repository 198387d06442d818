"""A seeded random program over hehub's one-ciphertext-per-call interface (examples/random_program.cpp: add, sub, mult, the tensor
product and its relinearisation apart, rotate, conjugate, rescale / mod_switch in place, plaintext operations, polynomial operators,
copies, moves, looks at words, results replacing operands) must print hehub's own digest however the layer runs it: one lane or
eight, eager or deferred, and at parity level A the same digest in every mode.  hehub's digests: tests/golden/random_program.json
(generated from hehub itself, tests/golden/make_random_program.py) and, where the prebuilt oracle/_ref/ref_randprog_cpu exists, that
program run on the spot."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_random_program import CASES, REF, digest  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "random_program.json")) as f:
    GOLDEN = json.load(f)["digests"]

REF_AMD = os.path.join(ROOT, "oracle", "_ref", "ref_randprog_amd")
MODES = {"1 lane": {"HEHUB_AMD_LANES": "1", "HEHUB_AMD_DEFER": "0"}, "8 lanes": {"HEHUB_AMD_LANES": "8", "HEHUB_AMD_DEFER": "0"},
         "deferred": {"HEHUB_AMD_DEFER": "1"}, "deferred, 8 lanes": {"HEHUB_AMD_DEFER": "1", "HEHUB_AMD_LANES": "8"},
         "default": {}}     # (the layer's default since round 6: recorded, four lanes)


def binary():
    from hehub_amd.build import build_example

    return build_example("random_program")


def test_every_case_has_hehubs_digest():
    assert set(GOLDEN) == {" ".join(str(a) for a in c) for c in CASES}
    if os.path.exists(REF):      # hehub itself, here and now
        c = CASES[0]
        assert digest(REF, c)[0] == GOLDEN[" ".join(str(a) for a in c)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(a) for a in c))
def test_random_program_prints_hehubs_digest_in_every_mode(case):
    want = GOLDEN[" ".join(str(a) for a in case)]
    for mode, env in MODES.items():
        got, text = digest(binary(), case, env)
        assert got == want, (mode, case, got, want, text)
    if os.path.exists(REF_AMD) and case[0] <= 12:   # hehub's own objects over the binding: every call crosses PCIe, small rings only
        assert digest(REF_AMD, case)[0] == want, ("binding", case)
    for mode in ("deferred", "default"):
        text = digest(binary(), case, MODES[mode])[1]
        assert "deferred 1" in text and "deferred_calls 0" not in text      # the calls really were recorded
    assert "deferred 0" in digest(binary(), case, MODES["1 lane"])[1]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [CASES[0], CASES[8], CASES[-4], CASES[-1]], ids=lambda c: "-".join(str(a) for a in c))
def test_random_program_at_parity_level_a(case):
    """canonical residues: one representative per word, so every mode agrees with every other (and differs from hehub's lazy words)"""
    got = {mode: digest(binary(), case, dict(env, HP_PARITY_LEVEL="A"))[0] for mode, env in MODES.items()}
    assert len(set(got.values())) == 1, got
