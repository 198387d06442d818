"""benchkit/host.py: how many single-threaded CPU-baseline processes this host can really run at once (VERDICT r02 item 8:
the P of cpu_baseline_node must respect the cgroup CPU quota, not only the affinity mask)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchkit import host  # noqa: E402


def _cg(tmp_path, **files):
    for name, text in files.items():
        p = tmp_path / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    return str(tmp_path)


def test_cpuset_lists():
    assert host.cpuset_count("0-63,128-191") == 128 and host.cpuset_count("5") == 1 and host.cpuset_count("0,2,4-5\n") == 4
    assert host.cpuset_count("") == 0 and host.cpuset_count(None) == 0


def test_quota_v2_and_v1(tmp_path):
    assert host.cpu_quota_cores(_cg(tmp_path / "a", **{"cpu.max": "1500000 100000\n"})) == 15.0
    assert host.cpu_quota_cores(_cg(tmp_path / "b", **{"cpu.max": "max 100000\n"})) is None
    assert host.cpu_quota_cores(_cg(tmp_path / "c", **{"cpu/cpu.cfs_quota_us": "250000", "cpu/cpu.cfs_period_us": "100000"})) == 2.5
    assert host.cpu_quota_cores(_cg(tmp_path / "d", **{"cpu/cpu.cfs_quota_us": "-1", "cpu/cpu.cfs_period_us": "100000"})) is None
    assert host.cpu_quota_cores(str(tmp_path / "missing")) is None


def test_usable_cores_takes_the_minimum(tmp_path):
    aff = len(os.sched_getaffinity(0))
    p, facts = host.usable_cores(64, _cg(tmp_path / "a", **{"cpu.max": "300000 100000", "cpuset.cpus.effective": "0-255"}))
    assert p == min(3, aff) and facts["cpu_quota_cores"] == 3.0 and facts["cpuset_cores"] == 256 and facts["cores_visible"] == aff
    p, _ = host.usable_cores(64, _cg(tmp_path / "b", **{"cpu.max": "50000 100000"}))     # half a core still runs one process
    assert p == 1
    p, facts = host.usable_cores(2, str(tmp_path / "none"))
    assert p == min(2, aff) and facts["cpu_quota_cores"] is None


def test_roofline_arithmetic():
    """benchkit/timing.py: the `roofline` object of the line is bytes / launch time against the 8 TB/s spec peak, with the
    per-limb PMC traffic of the kernel shape scaled to the launch"""
    from benchkit.timing import HBM_PEAK_GBS, rate_entry, roofline_entry

    n, limbs, steps = 1 << 15, 25600, 20
    alg = 16.0 * n * limbs                      # algorithmic bytes per step (one launch per step)
    r = roofline_entry("ntt", alg, steps, launches=steps, kern_ms=steps * 3.75, elapsed=steps * 8.6e-3, logn=15, spread=True, sclk_mhz=2200.0)
    # the digit-spread launch: VALUBusy 0.83 at a quarter of the HBM peak in the committed counters -> the line says "valu"; achieved /
    # peak / frac stay the HBM roofline (priced_against), `alu` prices the same launch against the issue peak of 1024 SIMDs
    assert r["bound"] == "valu" and r["priced_against"] == "hbm" and r["traffic_frac_of_hbm_peak"] < 0.5
    assert r["peak"] == HBM_PEAK_GBS == 8000.0 and r["unit"] == "GB/s" and r["launches"] == steps
    alu = r["alu"]
    assert alu["waves"] == limbs * 16 and alu["sclk_MHz"] == 2200.0 and 4000 < alu["valu_insts_per_wave"] < 5000
    assert abs(alu["frac_of_issue_peak"] - alu["valu_insts_per_wave"] * alu["waves"] * alu["cycles_per_inst_mix"] / (1024 * 2.2e9 * 3.75e-3)) < 1e-9
    assert 0.5 < alu["frac_of_issue_peak"] < 1.0
    assert abs(r["achieved"] - alg / 3.75e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert abs(r["share_of_step_time"] - 3.75 / 8.6) < 1e-9 and r["algorithmic_bytes_per_launch"] == alg
    # the committed PMC measurement of the digit-spread launch: less HBM traffic than algorithmic bytes (sources from L2, packed rows)
    assert r["traffic"] is not None and 0.3 * alg < r["traffic"] < alg and 0.5 < r["valu_busy"] < 1.0
    # a smaller ring has fewer waves per limb (32 coefficients per thread: N / 2048 waves): C2's launch of 4096 limbs at N = 16384
    c2 = roofline_entry("ntt", 16.0 * (1 << 14) * 4096, 20, launches=20, kern_ms=20 * 0.344, elapsed=20 * 0.35e-3, logn=14, spread=False, sclk_mhz=2250.0)
    assert c2["alu"]["waves"] == 4096 * 8 and 0.5 < c2["alu"]["frac_of_issue_peak"] < 1.0
    plain = roofline_entry("ntt", alg, steps, steps, steps * 3.9, steps * 3.9e-3, 15, spread=False)
    assert plain["traffic"] > r["traffic"] and abs(plain["traffic"] / alg - 1.0) < 0.05        # in-place launch: ~1.0 x algorithmic
    assert roofline_entry("elem", 24.0 * n * 2040, 5, 5, 5 * 0.28, 5 * 0.28e-3, 15, False)["bound"] == "hbm"   # no counters say otherwise
    assert roofline_entry("elem", 24.0 * n * 2040, 5, 5, 5 * 0.28, 5 * 0.28e-3, 15, False)["traffic"] is None
    e = rate_entry(units_per_launch=5120, bytes_per_unit=16.0 * n, steps=10, world=2, dt=10e-3, launches=10, kern_ms=9.0)
    assert abs(e["per_s"] - 5120 * 2 * 10 / 10e-3) < 1e-6 and abs(e["avg_launch_ms"] - 0.9) < 1e-12
    assert abs(e["frac_of_hbm_peak"] - 5120 * 16.0 * n / 0.9e-3 / 1e9 / 8000.0) < 1e-12
    assert "achieved_GBps" not in rate_entry(1, 1.0, 1, 1, 1.0, 0, 0.0)


def test_rank_placement_on_the_gpus_numa_node():
    """benchkit/host.py: a bench rank binds itself to the CPUs of its GPU's NUMA node (sysfs lookup injected here); a platform that
    does not say (-1 / empty) or HP_BENCH_NO_AFFINITY leaves the affinity mask alone"""
    import os

    from benchkit.host import bind_to_gpu_numa, parse_cpulist

    assert parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11} and parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    try:
        first = min(before)
        r = bind_to_gpu_numa(0, lookup=lambda d: (1, f"{first}"))
        assert r["numa_node"] == 1 and r["cpus_bound"] == 1 and os.sched_getaffinity(0) == {first}
        os.sched_setaffinity(0, before)
        assert bind_to_gpu_numa(0, lookup=lambda d: (-1, ""))["cpus_bound"] == 0 and os.sched_getaffinity(0) == before
        assert bind_to_gpu_numa(0, lookup=lambda d: (0, "100000-100001"))["cpus_bound"] == 0   # CPUs this process may not use: nothing bound
        os.environ["HP_BENCH_NO_AFFINITY"] = "1"
        assert "skipped" in bind_to_gpu_numa(0, lookup=lambda d: (0, f"{first}")) and os.sched_getaffinity(0) == before
    finally:
        os.environ.pop("HP_BENCH_NO_AFFINITY", None)
        os.sched_setaffinity(0, before)


def test_contract_line_is_compact_and_strict(tmp_path, monkeypatch):
    """benchkit/line.py on round 5's 34.5 KB line (the one the driver could not read): the LAST stdout line is the contract, < 8 KB,
    strict JSON; the sections travel as `#section` lines and in the side file; collect() puts them back together."""
    import io
    import json

    from benchkit import line as L

    with open(os.path.join(ROOT, "profiles", "r05y_bench_default.json")) as f:
        res = json.load(f)
    res["level_a"]["poison"] = float("nan")            # a non-finite float anywhere must come out as null, never as NaN
    res["pipeline_roofline"]["inf"] = float("inf")
    monkeypatch.chdir(tmp_path)
    buf = io.StringIO()
    text = L.emit(res, buf)
    lines = buf.getvalue().splitlines()
    assert lines[-1] == text and len(text.encode()) < L.LINE_LIMIT < 8193 and text.startswith("{")
    assert len([l for l in lines if l.startswith("{")]) == 1 and all(l.startswith(L.PREFIX) for l in lines[:-1])
    line = L.strict_loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "pipeline_roofline", "cpu_baseline_node", "cpu_model", "parity_level", "verified", "summary"):
        assert k in line, k
    assert line["pipeline_roofline"]["inf"] is None and line["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert abs(line["value"] - res["value"]) < 1e-5 * res["value"]
    sm = line["summary"]
    assert sm["ntt_fwd_per_s"] == pytest.approx(res["ntt"]["steady_state"]["forward"]["per_s"], rel=1e-5) and sm["c2_fwd_frac"] == pytest.approx(res["c2"]["forward"]["roofline"]["frac"], rel=1e-5) and sm["level_a_per_s"] > 3.5e4 and sm["object_api_batched_per_s"] > 2e4
    got, full = L.collect(buf.getvalue())
    assert got == line and full["level_a"]["poison"] is None and set(line["sections"]["names"]) <= set(full)
    assert len(buf.getvalue().encode()) < L.STDOUT_LIMIT          # everything printed: ~21 KB of the 34.5 KB the sections hold
    # stdout carries BRIEF copies (5 digits, strings cut, nothing deeper than three levels); the side file has every digit and level
    assert full["ntt"]["by_N"]["4096"]["forward"] == L.DEEPER
    assert full["ntt"]["steady_state"]["forward"]["per_s"] == pytest.approx(res["ntt"]["steady_state"]["forward"]["per_s"], rel=1e-4)
    with open(tmp_path / "bench_sections.json") as f:
        side = L.strict_loads(f.read())
    assert side["ntt"]["by_N"]["4096"]["forward"]["per_s"] == pytest.approx(res["ntt"]["by_N"]["4096"]["forward"]["per_s"], rel=1e-8)
    for name in line["sections"]["names"]:
        assert L.brief(side[name]) == full[name], name
    with pytest.raises(ValueError):
        L.strict_loads('{"a": NaN}')
    with pytest.raises(ValueError):     # a line over the limit is refused by the reader the tests use
        L.collect("x\n{" + '"a": "' + "y" * 9000 + '"}')
