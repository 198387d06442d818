"""Everything that compares a MEASURED rate or time with a threshold or with another measurement (`pytest -m perf`, on an MI355X).

Not part of the correctness tiers (`-m gpu` / `-m "not gpu"` never select these: tests/conftest.py): on a shared or noisy box a
timing can miss without anything being wrong, and with `pytest -x` a miss would hide every later parity test.  The bounds are loose
ratios; the typical values are in the comments and in profiles/."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

pytestmark = pytest.mark.perf


def need_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.fixture(scope="module")
def default_line(tmp_path_factory):
    need_gpu()
    from test_bench_contract import run_bench

    line, full, _ = run_bench(["--steps", "10", "--warmup", "3", "--cpu-seconds", "2", "--cpu-procs", "2", "--cpu-node-seconds", "2"], timeout=1800,
                              cwd=str(tmp_path_factory.mktemp("bench")))
    return line, full


def test_headline_and_rooflines(default_line):
    r, full = default_line
    assert r["value"] > 25000 and 0.35 < r["roofline"]["frac"] < 0.6                        # typically 29.2 - 29.7 k, 0.43 - 0.44
    assert 0.5 < r["roofline"]["alu"]["frac_of_issue_peak"] < 1.1 and r["roofline"]["alu"]["sclk_MHz"] > 500
    assert 0.6 < r["pipeline_roofline"]["frac_of_hbm_peak"] < 1.0
    ntt = full["ntt"]
    assert ntt["steady_state"]["forward"]["frac_of_hbm_peak"] >= ntt["forward"]["frac_of_hbm_peak"] * 0.95
    for ent in ntt["by_N"].values():
        for d in ("forward", "inverse"):
            assert 0.05 < ent[d]["frac_of_hbm_peak"] < 1
    for ent in full["ckks_by_N"].values():
        assert 0.3 < ent["A_step_frac_of_hbm_peak"] < 1
    assert full["ckks_by_N"]["4096"]["per_s"] > 4 * full["ckks_by_N"]["32768"]["per_s"]
    for op in ("mul", "add"):
        assert 0.2 < full["coeffwise"][op]["frac_of_hbm_peak"] < 1
    c2 = full["c2"]
    for d in ("forward", "inverse"):
        assert 0.3 < c2[d]["roofline"]["frac"] < 1                                          # typically 0.48 / 0.46
        alu = c2[d]["roofline"].get("alu")
        if alu and "frac_of_issue_peak" in alu:
            assert 0 < alu["frac_of_issue_peak"] < 1.1
    bgv = full["bgv"]
    assert 0.3 < bgv["pipeline_roofline"]["frac_of_hbm_peak"] < 1 and bgv["per_s"] > 2 * r["value"]
    assert 2000 < full["hbm_copy_ceiling_GBps"] < 8000 and full["hbm_stream_ceiling_GBps"] >= full["hbm_copy_ceiling_GBps"]
    assert all(2000 < v < 8000 for v in full["hbm_copy"]["stream_mix_GBps"].values())
    chip = full["chip"]
    assert chip["timed_region"]["samples"] >= 3 and 500 < chip["timed_region"]["sclk_MHz"] <= 2500 and chip["timed_region"]["socket_power_W"] > 100


def test_cpu_baselines_scale(default_line):
    r, _ = default_line
    node = r["cpu_baseline_node"]
    assert node["value"] > r["cpu_baseline"]["value"] * 0.8 and 0 < node["parallel_efficiency"] < 1.5


def test_step_accounting(default_line):
    r, full = default_line
    step = full["step"]
    assert 0.8 * step["wall_ms_per_step"] < step["kernel_ms_per_step"] < 1.1 * step["wall_ms_per_step"]
    tr = step["step_traffic"]
    assert 0.2 < tr["measured_over_A_step"] < 1.0 and tr["measured_over_A_min"] > 1.0 and 0.1 < tr["frac_of_hbm_peak"] < 1.0
    assert all(0 < k["measured_frac_of_hbm_peak"] < 1.0 for k in step["kernels"].values() if "measured_frac_of_hbm_peak" in k)


def test_level_a_is_faster(default_line):
    r, full = default_line
    la = full["level_a"]
    for k in ("ckks", "bgv"):
        assert la[k]["speedup_vs_level_b"] > 1.0
    assert la["ckks"]["roofline"]["frac"] > r["roofline"]["frac"]
    for key in ("32768", "steady_32768", "c2"):
        assert la["ntt"][key]["forward"]["frac_of_hbm_peak"] > 0.3 and la["ntt"][key]["inverse"]["frac_of_hbm_peak"] > 0.3
    assert la["ntt"]["steady_32768"]["forward"]["frac_of_hbm_peak"] > full["ntt"]["steady_state"]["forward"]["frac_of_hbm_peak"]


def test_object_api_rates(default_line):
    """hehub's object API at C3 (typically: eager single calls 12 - 13 k, batched form 27 - 29 k, the unchanged loop 25 - 27 k hom-mult/s)"""
    _, full = default_line
    api = full["object_api"]
    assert api["batched_call"]["per_s"] > 15000 and api["batched_call"]["per_s"] > 1.4 * api["single_calls"]["per_s"]
    assert api["unchanged_loop"]["per_s"] > 20000 and api["unchanged_loop"]["per_s"] > 1.3 * api["single_calls"]["per_s"]
    assert api["independent_chains"]["speedup"] > 1.1
    mv = api["matvec"]
    assert mv["ms"]["deferred"] < 1.1 * mv["ms"]["eager"]


def test_c3_single_calls_batched_form_and_lanes():
    need_gpu()
    from test_object_api import build_example, run

    eager = run(build_example(), [15, 10, 64, "serial", 3], {"HEHUB_AMD_DEFER": "0"})
    lazy = run(build_example(), [15, 10, 64, "serial", 3], {"HEHUB_AMD_DEFER": "1"})
    assert lazy["serial_per_s"] > 12000 and lazy["serial_per_s"] > 1.25 * eager["serial_per_s"], (lazy, eager)   # (typically 22 k against 12.5 k)
    r = run(build_example(), [15, 10, 64, "all", 3, 8, 8, 4], {"HEHUB_AMD_DEFER": "0"})
    assert r["batch_per_s"] > 15000 and r["batch_per_s"] > 1.3 * r["serial_per_s"], r
    assert r["chain_ms"][1] < 1.1 * r["chain_ms"][0], r   # (typically 0.7)


def test_matvec_recorded_rotations_do_not_lose():
    need_gpu()
    from make_matvec import CASES, run
    from test_matvec import binary

    def timed(case, ok):   # (a miss is measured again once before it counts)
        for _ in range(2):
            got, ms, text = run(binary(), case, reps=4)
            if ok(ms):
                break
        assert ok(ms), (ms, text)

    timed((13, 6, 16, "short"), lambda ms: ms["deferred"] < 0.6 * ms["eager"])                     # typically 0.25
    timed(CASES[6], lambda ms: ms["deferred"] < 1.05 * ms["eager"] and ms["batched-form"] < 1.1 * ms["eager"])   # typically 0.8 - 0.9


def test_resident_chain_latency():
    """C3 shape, one ciphertext at a time through hehub's API, call by call: device latency, not PCIe (was 10.8 ms per operation with staged
    operands; hp_dev_* at batch 1 is ~0.12 ms).  The 23 timed mult + add iterations also carry the first use of three of the four lanes
    (a stream, a workspace: ~7 ms each, once per process): ~0.14 + 21 / 23 ms"""
    need_gpu()
    from test_host_residency import build_chain, run

    own = run(build_chain(), (15, 10, 24), {"HEHUB_AMD_DEFER": "0"})
    assert own["mult+add"] < 2.5 and own["rotate"] < 1.0, own
