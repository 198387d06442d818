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
