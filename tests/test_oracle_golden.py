"""The oracle against the committed golden vectors (tests/golden/golden.json,
produced from the unmodified reference by tests/golden/make_golden.py).  This is
the pin that travels to the GPU box, where /root/reference does not exist."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)["cases"]


def test_oracle_matches_golden(orc, golden):
    from cases import run_cases

    got = run_cases(orc, big=True)
    assert set(got) == set(golden)
    bad = [k for k in sorted(golden) if got[k] != golden[k]]
    assert not bad, f"{len(bad)} cases differ from the reference-generated fixtures: {bad[:5]}"


def test_survey_known_answers(orc, golden):
    """Known answers quoted in SURVEY.md section 8a (measured on the compiled reference)."""
    assert golden["psi_q576460752272228353_logn12"]["value"] == 41473362160949302
    assert golden["psi_q1099510054913_logn14"]["value"] == 81696219706
    assert golden["psi_q65537_logn4"]["value"] == 65529
    assert golden["harvey_scalar"]["value"] == 816624806244
    assert golden["inverse_65537_mod_q40"]["value"] == 395197504517
    assert orc.unity_root(576460752272228353, 4096) == 41473362160949302


def test_plain_numpy_splitmix_is_the_same_stream():
    """tests/splitmix.py (used by the fixture cases and by the engine-vs-fixture test) == the generator of the oracle module"""
    import numpy as np
    from oracle.pyoracle import SplitMix as A
    from splitmix import SplitMix as B

    a, b = A(42), B(42)
    assert (a.words(1000, 0) == b.words(1000, 0)).all() and (a.words(77, 65537) == b.words(77, 65537)).all()
    assert (a.poly((2, 3, 64), [97, 193, 257]) == b.poly((2, 3, 64), [97, 193, 257])).all() and a.state == b.state
    # first word of the seed-42 stream reduced mod q is what the known answers were made from
    assert int(B(0).words(1)[0]) == 0xE220A8397B1DCDAF


def test_golden_case_list_runs_without_scalar_entry_points(orc):
    """an implementation without the scalar helpers (the engine adapter) skips only the scalar known answers"""
    from cases import run_cases

    class NoScalars:
        def __getattr__(self, name):
            if name in ("mul_mod_harvey_lazy", "inverse_mod_prime", "unity_root"):
                raise AttributeError(name)
            return getattr(orc, name)

    got = run_cases(NoScalars(), big=False, fnv=orc.fnv)
    full = run_cases(orc, big=False)
    assert set(full) - set(got) == {k for k in full if k.startswith(("psi_", "harvey_scalar", "inverse_"))}
    assert all(got[k] == full[k] for k in got)
