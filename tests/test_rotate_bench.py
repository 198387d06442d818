"""hehub's OWN benchmark (bench/benchmarks.cpp:21-37: ckks::rotate(ct, rot_key, 1), one ciphertext per call, N = 2^12 .. 2^15 with
the modulus chains of ckks::create_params(N, scaling_bits) -- 36- to 55-bit moduli, up to 15 of them) as examples/rotate_bench.cpp on
synthetic words: the MI355X layer must print hehub's digests (tests/golden/rotate_bench.json, generated from hehub itself by
tests/golden/make_rotate_bench.py; where the prebuilt oracle/_ref/ref_rotbench_cpu exists, also that program run on the spot) however
it runs the calls -- one lane, four lanes, recorded."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_rotate_bench import LOGNS, REF, run  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "rotate_bench.json")) as f:
    GOLDEN = {int(k): v for k, v in json.load(f)["digests"].items()}


def binary():
    from hehub_amd.build import build_example

    return build_example("rotate_bench")


def test_every_parameter_set_has_hehubs_digest():
    assert set(GOLDEN) == set(LOGNS)
    if os.path.exists(REF):      # hehub itself, here and now (the small sets: the big ones take seconds per rotation on one core)
        rows, _ = run(REF, 1, 12)
        assert rows[12][0] == GOLDEN[12]
        rows, _ = run(REF, 1, 13)
        assert rows[13][0] == GOLDEN[13]


def test_example_builds():
    assert os.path.exists(binary())


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"HEHUB_AMD_DEFER": "0"}, {"HEHUB_AMD_LANES": "1", "HEHUB_AMD_DEFER": "0"}, {"HEHUB_AMD_DEFER": "1"}, {},
                                 {"HP_SPLIT_MAX_ITEMS": "0", "HEHUB_AMD_DEFER": "0"}],
                         ids=["lanes", "one-lane", "deferred", "default", "tiled-transforms-only"])
def test_rotate_bench_prints_hehubs_digests(env):
    rows, text = run(binary(), 3, 0, env)
    assert set(rows) == set(LOGNS), text
    for logn in LOGNS:
        assert rows[logn][0] == GOLDEN[logn], (logn, env, rows[logn], GOLDEN[logn], text)
    if env.get("HEHUB_AMD_DEFER", "1") == "1":     # (nothing in the environment: recorded, the default)
        assert "deferred 1" in text and "deferred_calls 0" not in text, text
