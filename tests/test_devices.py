"""hehub's object API over SEVERAL device ranks (hehub_amd/host: "devices"; SURVEY.md 8e: independent ciphertexts
shard across the GPUs of a node, contiguous slices, no collective).  hehub has no devices -- nothing to cite there; the contract is
that the words do not depend on where a call ran: every program must print hehub's own digests (tests/golden/*.json, generated from
hehub on the CPU; oracle/_ref/ref_*_cpu on the spot where it travelled) with 2 / 3 / 8 ranks, call by call and recorded, and the
layer's own counters must say that every rank really worked.  On a one-GPU box the ranks share GPU 0 (HEHUB_AMD_DEVICES=0,0,..: one
engine family, stream set, block pool, key cache and table set per rank all the same)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

RANKS = {2: "0,0", 3: "0,0,0", 8: "0,0,0,0,0,0,0,0"}
EAGER = {"HEHUB_AMD_DEFER": "0"}


def per_rank_calls(text):
    m = re.search(r"devices (\d+) engine calls per device rank:((?: \d+)+); copies between ranks (\d+)", text)
    assert m, text[-1500:]
    return int(m.group(1)), [int(x) for x in m.group(2).split()], int(m.group(3))


def run_text(binary, args, env):
    base = {k: v for k, v in os.environ.items() if k not in ("HEHUB_AMD_DEFER", "HEHUB_AMD_DEVICES", "HEHUB_AMD_LANES")}
    out = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, timeout=1800, env=dict(base, **env))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    return out.stdout


def digests(text):
    return {m.group(1): m.group(2) for m in re.finditer(r"^([\w-]+) digest (\w+)", text, re.M)}


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("shape", [(12, 4, 6), (13, 6, 9)])
def test_independent_mults_over_device_ranks(shape, ranks):
    """B independent ckks::mult + rescale_inplace, the batched forms, independent chains over lanes: every mode, call by call and
    recorded, prints the digests of ONE rank -- and of hehub itself on the CPU"""
    from test_object_api import REF_CPU, build_example

    args = list(shape) + ["all", 2, 8, 3, 2]
    one = digests(run_text(build_example(), args, EAGER))
    assert set(one) >= {"serial", "batch", "serial-chain", "batch-chain", "chains", "chains-lanes"}
    if os.path.exists(REF_CPU):
        ref = digests(run_text(REF_CPU, list(shape) + ["all", 1, 8, 3, 2], {}))
        for k in ("serial", "serial-chain", "chains"):
            assert ref[k] == one[k], (k, ref, one)
    for mode in (EAGER, {}, {"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "1"}):
        text = run_text(build_example(), args, dict(mode, HEHUB_AMD_DEVICES=RANKS[ranks]))
        assert digests(text) == one, (mode, ranks, digests(text), one)
        n, calls, _ = per_rank_calls(text)
        assert n == ranks and len(calls) == ranks
        # B = 6 or 9 independent ciphertexts and 3 chains: with 2 or 3 ranks every rank has work in every mode
        if ranks <= 3:
            assert all(c > 0 for c in calls), (mode, calls)
        else:
            assert sum(1 for c in calls if c > 0) >= 3, (mode, calls)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_random_program_over_device_ranks(ranks):
    """the seeded random program over hehub's whole interface (operands re-used across calls, results replacing operands, copies, looks):
    hehub's digest wherever the calls ran; operands that meet from different ranks are copied over (peer copies counted)"""
    from make_random_program import CASES, digest
    from test_random_program import GOLDEN, binary

    saw_copies = False
    for case in (CASES[0], CASES[3], CASES[8], CASES[-1]):
        want = GOLDEN[" ".join(str(a) for a in case)]
        for mode in ({"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "1"}, {"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "8"}, {"HEHUB_AMD_DEFER": "1"}):
            got, text = digest(binary(), case, dict(mode, HEHUB_AMD_DEVICES=RANKS[ranks]))
            assert got == want, (case, mode, ranks, got, want, text[-1500:])
            n, calls, copies = per_rank_calls(text)
            assert n == ranks and sum(1 for c in calls if c > 0) >= 2, (case, mode, calls)
            saw_copies = saw_copies or copies > 0
    assert saw_copies       # (a binary operation on two results that were made on different ranks: one of them moved)


@pytest.mark.gpu
def test_random_program_over_device_ranks_at_level_a():
    from make_random_program import CASES, digest
    from test_random_program import binary

    for case in (CASES[0], CASES[8]):
        got = {(r, d): digest(binary(), case, {"HP_PARITY_LEVEL": "A", "HEHUB_AMD_DEFER": d, "HEHUB_AMD_DEVICES": RANKS.get(r, "0")})[0]
               for r in (1, 3) for d in ("0", "1")}
        assert len(set(got.values())) == 1, got


@pytest.mark.gpu
def test_matvec_and_rotate_bench_over_two_ranks():
    """hehub's circuit-level caller (one vector under 30 keys: ONE operand -- the rotations stay on its device) and hehub's own benchmark
    loop (dependent on nothing: consecutive rotations of the same ciphertext stay where it lives)"""
    from make_matvec import CASES, run as run_matvec
    from make_rotate_bench import LOGNS, run as run_rot
    from test_matvec import GOLDEN as MV, binary as mv_binary, key
    from test_rotate_bench import GOLDEN as RB, binary as rb_binary

    case = CASES[2]     # N = 8192, L = 6, width 20: 38 rotations of one vector under 38 keys
    got, _, text = run_matvec(mv_binary(), case, {"HEHUB_AMD_DEVICES": "0,0"}, reps=2)
    assert got["eager"] == got["deferred"] == got["batched-form"] == MV[key(case)], text
    rows, text = run_rot(rb_binary(), 3, 0, {"HEHUB_AMD_DEVICES": "0,0"})
    for logn in LOGNS:
        assert rows[logn][0] == RB[logn], (logn, text)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"HEHUB_AMD_DEVICES": "0,0,0"}, {"HEHUB_AMD_DEVICES": "0,0", "HEHUB_AMD_DEFER": "0"}], ids=["3-ranks-recorded", "2-ranks-call-by-call"])
def test_reference_unit_tests_over_device_ranks(env):
    """hehub's unit tests for this path (tests/cpp/host_api_test.cpp) with the layer spread over device ranks"""
    from test_host_api import build_binary

    base = {k: v for k, v in os.environ.items() if k not in ("HEHUB_AMD_DEFER", "HEHUB_AMD_LANES")}
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=900, env=dict(base, **env))
    assert out.returncode == 0 and "All tests passed" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


@pytest.mark.gpu
def test_recorded_calls_over_ranks_while_another_lane_has_an_open_call():
    """Regression (round 6, found by tools/fuzz_random_program.py after 129 programs): the queue may run in the MIDDLE of a call on another
    slot (a look, an in-place operator on a recorded result); its groups switch to lane 0 of their rank and may order themselves behind the
    outer call's slot -- whose ticket is still open.  What the outer call enqueued afterwards then passed for `seen`, and a block it still
    used looked free to the other slot: 15 runs of 24 printed another digest (2 ranks, recorded, 4 lanes, level A; 3 of 24 at level B).
    The outer call now continues under a new ticket (layer.hpp: ~OpScope).  The same seed, many times: one digest, hehub's."""
    from make_random_program import REF, digest
    from test_random_program import binary

    case = (12, 4, 12, 1085, 41028, 0)
    want_b = digest(binary(), case, {"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "1"})[0]
    if os.path.exists(REF):
        assert digest(REF, case)[0] == want_b           # hehub itself on the CPU
    want_a = digest(binary(), case, {"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "1", "HP_PARITY_LEVEL": "A"})[0]
    for env, want in (({"HP_PARITY_LEVEL": "A", "HEHUB_AMD_DEVICES": "0,0"}, want_a), ({"HP_PARITY_LEVEL": "A", "HEHUB_AMD_DEVICES": "0,0", "HP_SPLIT_MAX_ITEMS": "0"}, want_a),
                      ({"HEHUB_AMD_DEVICES": "0,0"}, want_b), ({"HEHUB_AMD_DEVICES": "0,0,0", "HEHUB_AMD_LANES": "8"}, want_b), ({"HEHUB_AMD_LANES": "8"}, want_b)):
        got = [digest(binary(), case, env)[0] for _ in range(10)]
        assert got == [want] * 10, (env, got, want)
