"""bench.py's LAST stdout line is ONE compact JSON line (< 8 KB, strict JSON) carrying every field of the driver's contract; the
detailed sections travel as `#section` lines before it and in bench_sections.json (benchkit/line.py).  Structure only here: fields,
digests, verification verdicts.  Everything that compares a measured rate with a number or with another rate lives in
tests/test_perf.py (`-m perf`, not part of the correctness tier)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit.line import LINE_LIMIT, STDOUT_LIMIT, brief, collect, strict_loads   # noqa: E402

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def run_bench(flags, timeout=900, env=None, cwd=None):
    """bench.py as the driver runs it; returns (the last stdout line parsed ALONE and strictly, contract + sections merged, process).
    The sections come from the side file when the run wrote one into `cwd` (every digit, every level), else from the brief copies on
    stdout."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, capture_output=True, text=True, timeout=timeout,
                         env=env, cwd=cwd)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = out.stdout.splitlines()
    assert len([l for l in lines if l.startswith("{")]) == 1, out.stdout[-2000:]     # ONE JSON line ...
    last = lines[-1]
    assert last.startswith("{") and len(last.encode()) < LINE_LIMIT, len(last)        # ... the last one, compact
    line = strict_loads(last)                                                         # ... strict JSON: no NaN / Infinity
    got, full = collect(out.stdout)
    assert got == line
    assert len(out.stdout.encode()) < STDOUT_LIMIT, len(out.stdout)                   # ... and everything printed stays small
    side_path = os.path.join(cwd, "bench_sections.json") if cwd else None
    if side_path and os.path.exists(side_path) and line.get("sections", {}).get("file") == "bench_sections.json":
        with open(side_path) as f:
            side = strict_loads(f.read())
        for name in line["sections"]["names"]:     # the stdout lines are the brief copies of what the side file holds
            assert json.dumps(brief(side[name]), sort_keys=True) == json.dumps(full[name], sort_keys=True), name
        full = dict(line, **{k: side[k] for k in line["sections"]["names"]})
    return line, full, out


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ckks", "ntt"])
def test_bench_line_has_the_contract_fields(workload):
    r, full, _ = run_bench(["--workload", workload, "--batch", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--cpu-procs", "0"])
    assert REQUIRED <= set(r), REQUIRED - set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["dtype"] == "u64" and r["data"] == "synthetic" and r["vs_baseline"] is None and r["scaling"] == "weak"
    assert "workload" in r["config"] and r["value"] > 0 and r["ms_per_step"] > 0
    roof = r["roofline"]
    # achieved / peak / frac are the HBM roofline on algorithmic bytes (priced_against); `bound` names what the committed counters say
    # limits the kernel: "valu" for the transforms (VALUBusy >= 0.6 at < 0.5 of the HBM peak in real traffic; then `alu` prices the
    # launch against the issue peak), "hbm" otherwise
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] in ("hbm", "valu") and roof["peak"] == 8000.0
    assert roof["priced_against"] == "hbm" and roof["unit"] == "GB/s"
    if roof["bound"] == "valu":
        assert roof["valu_busy"] >= 0.6 and "traffic_frac_of_hbm_peak" in roof and roof["alu"]["valu_insts_per_wave"] > 1000   # (committed counters)
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5 * roof["frac"] and 0 < roof["frac"] < 1
    cpu = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] in ("reference", "port") and cpu["cores"] == 1
    # every output of the timed buffers was compared with the CPU checker (periodic batch: 3 checker evaluations)
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == 8 and r["verify"]["checker_evaluations"] == 3
    # a one-rank run creates the RCCL communicator too (what an N-GPU job does first), and says so
    assert r["dist_ranks"] == 1 and r["rccl_ranks"] == 1 and r["rccl"]["initialised"] is True and r["rccl"]["backend"] == "nccl"
    assert "chip" in full and "placement" in full and r["sections"]["stdout_prefix"] == "#section"


@pytest.mark.gpu
def test_default_line_carries_both_halves_of_the_metric(tmp_path):
    """The default command (C3, batch 256) with a short CPU budget.  The compact line: hom-mult/s with roofline + verification + CPU
    baselines + one number per section; the sections (side file == `#section` lines): the limb-transform rates at N = 4096..32768, the
    coefficient-wise rates, BASELINE configs 2 and 5 at their exact shapes, level A, hehub's object API -- every one verified."""
    r, full, _ = run_bench(["--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--cpu-procs", "2", "--cpu-node-seconds", "1"],
                           timeout=1800, cwd=str(tmp_path))
    assert REQUIRED <= set(r) and r["metric"] == "ckks_hom_mult_per_s" and r["config"]["batch_per_gpu"] == 256
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == 256 and r["verify"]["checker"] in ("reference", "port")
    # (run_bench compared the brief stdout copies with the side file and handed back the side file's sections)
    assert r["sections"]["file"] == "bench_sections.json" and os.path.exists(tmp_path / "bench_sections.json")
    sm = r["summary"]
    assert {"ntt_fwd_per_s", "ntt_inv_per_s", "ntt_fwd_frac", "c2_fwd_per_s", "c2_fwd_frac", "c2_inv_frac", "bgv_per_s", "level_a_per_s",
            "object_api_single_per_s", "object_api_batched_per_s", "object_api_unchanged_loop_per_s", "step_traffic_bytes_per_op",
            "hbm_copy_ceiling_GBps", "all_sections_verified"} <= set(sm), sm
    assert sm["all_sections_verified"] is True and sm["c2_verified_limbs"] == 4096 and all(v > 0 for v in sm.values() if isinstance(v, float))
    ntt = full["ntt"]
    assert ntt["N"] == 32768 and ntt["verified"] is True and set(ntt["by_N"]) == {"4096", "8192", "16384", "32768"}
    assert ntt["steady_state"]["limbs_per_launch"] == 25600 and ntt["steady_state"]["verified"] is True
    assert sm["ntt_fwd_per_s"] == pytest.approx(ntt["steady_state"]["forward"]["per_s"], rel=1e-5)
    for ent in ntt["by_N"].values():
        assert ent["verified"] is True
        for d in ("forward", "inverse"):
            assert ent[d]["per_s"] > 0 and 0 < ent[d]["frac_of_hbm_peak"] < 1 and ent[d]["achieved_GBps"] > 0
    assert set(full["ckks_by_N"]) == {"4096", "8192", "16384", "32768"}
    for ent in full["ckks_by_N"].values():
        assert ent["verified"] is True and ent["per_s"] > 0
    cw = full["coeffwise"]
    for op in ("mul", "add"):
        assert cw[op]["verified"] is True and 0 < cw[op]["frac_of_hbm_peak"] < 1
    node = r["cpu_baseline_node"]
    assert node["cores"] == 2 and node["value"] > 0 and "cores_visible" in node
    assert "cpu_quota_cores" in node and "host_limited" in node and node["parallel_efficiency"] > 0
    # BASELINE configs 2 and 5 at their exact shapes, and the measured stream ceiling
    c2 = full["c2"]
    assert c2["N"] == 16384 and c2["limbs"] == 4 and c2["batch_per_gpu"] == 1024 and c2["verified"] is True and c2["verified_limbs"] == 4096
    assert c2["moduli"] == [1125899904679937, 1125899903827969, 1125899903500289, 1125899903107073]
    for d in ("forward", "inverse"):
        assert c2[d]["per_s"] > 0 and 0 < c2[d]["roofline"]["frac"] < 1 and c2[d]["cpu_baseline"]["value"] > 0
        alu = c2[d]["roofline"].get("alu")
        if alu:   # N / 2048 = 8 waves per limb at N = 16384
            assert alu["waves"] == 4096 * 8
    bgv = full["bgv"]
    assert bgv["N"] == 8192 and bgv["L"] == 6 and bgv["plain_modulus"] == 65537 and bgv["batch_per_gpu"] == 512
    assert bgv["verified"] is True and bgv["verified_outputs"] == 512 and bgv["A_step_bytes_per_op"] == 396 * 65536
    assert bgv["pipeline_roofline"]["frac_of_hbm_peak"] > 0 and 0 < bgv["roofline"]["frac"] < 1 and bgv["cpu_baseline"]["value"] > 0
    assert full["hbm_copy"]["engine_copy_verified"] is True and set(full["hbm_copy"]["stream_mix_GBps"]) == {"1R:1W", "2R:1W", "4R:3W"}
    chip = full["chip"]
    assert {"timed_region", "ntt", "c2", "coeffwise", "bgv", "level_a"} <= set(chip)
    assert isinstance(r["cpu_model"], str) and r["cpu_model"] != "unknown"
    for d in ("forward", "inverse"):
        assert ntt[d]["cpu_baseline"]["value"] > 0 and "N=32768" in ntt[d]["cpu_baseline"]["sample"]
    # round 5: the real bound of the dominant launch, every launch of a step with the measured traffic of the step, hehub's object API
    roof = r["roofline"]
    assert roof["bound"] == "valu" and roof["priced_against"] == "hbm" and roof["alu"]["waves"] == 25600 * 16
    step = full["step"]
    assert {"tensor", "intt", "ntt", "ks_inner", "ntt_drop"} <= set(step["kernels"])
    assert abs(sum(k["ms_per_step"] for k in step["kernels"].values()) - step["kernel_ms_per_step"]) < 1e-4 * step["kernel_ms_per_step"]
    a_step = sum(k.get("algorithmic_bytes_per_step", 0) for k in step["kernels"].values())
    assert abs(a_step - 256 * r["config"]["A_step_bytes_per_op"]) < 1e-3 * a_step          # the families' shares add up to A_step
    tr = step["step_traffic"]
    meas = [k["measured_bytes_per_step"] for k in step["kernels"].values() if "measured_bytes_per_step" in k]
    assert len(meas) >= 5 and abs(sum(meas) - 256 * tr["measured_bytes_per_op"]) < 0.02 * sum(meas)   # (committed counters, not timings)
    assert full["level_a"]["ckks"]["step"]["step_traffic"]["measured_bytes_per_op"] < tr["measured_bytes_per_op"]
    api = full["object_api"]
    assert api["verified"] is True and api["digests_equal"] is True and api["deferred"]["digests_equal_eager"] is True
    assert api["deferred"]["fused_triples"] >= 256 and api["unchanged_loop"]["fused_triples"] >= 256 and "numa_node" in full["placement"]
    assert api["unchanged_loop"]["environment"] == "none" and api["unchanged_loop"]["digest_equal"] is True
    dv = api["devices"]     # the same program over device ranks (here: two ranks sharing the GPU): the one-device digests, every rank worked
    assert dv["verified"] is True and dv["ranks"] >= 2 and len(dv["engine_calls_by_rank"]) == dv["ranks"] and min(dv["engine_calls_by_rank"]) > 0
    assert dv["ranks_share_one_gpu"] is False or dv["digests_equal_one_device"] is True
    assert sm["object_api_devices"]["ranks"] == dv["ranks"]
    mv = api["matvec"]      # the diagonal loop of matrix_vector_mul_short: every mode prints hehub's digest
    assert mv["verified"] is True and set(mv["digests"]) == {"eager", "deferred", "batched-form"}
    assert mv.get("cpu_reference_digest_equal", True) is True
    rb = api["reference_benchmark"]      # hehub's own benchmark (bench/benchmarks.cpp): hehub's digests at its four parameter sets
    assert rb["verified"] is True and set(rb["by_N"]) == {"4096", "8192", "16384", "32768"}
    assert all(e["digests_equal"] and e["look_after_every_call"] > 0 and e.get("cpu_reference_digest_equal", True) for e in rb["by_N"].values())
    la = full["level_a"]
    assert r["parity_level"] == "B"
    for k, outs in (("ckks", 256), ("bgv", 512)):
        assert la[k]["verified"] is True and la[k]["verified_outputs"] == outs
    for key in ("32768", "steady_32768", "c2"):     # the limb transforms as residues (hp_dev_ntt_residues / hp_dev_intt_residues)
        assert la["ntt"][key]["verified"] is True


@pytest.mark.gpu
def test_roofline_only_line_has_no_side_legs():
    r, full, _ = run_bench(["--batch", "8", "--steps", "2", "--warmup", "1", "--roofline-only"])
    assert "roofline" in r and not ({"ntt", "coeffwise", "cpu_baseline", "cpu_baseline_node", "verified"} & set(full))


@pytest.mark.gpu
def test_bench_refuses_two_gpus_on_a_one_gpu_box():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 2 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_two_ranks_end_to_end_on_one_gpu():
    """The N > 1 path of bench.py as the driver runs it (`bench.py --gpus 2` starts its own ranks; shard seeds per rank, fences,
    max over ranks, verification on every rank + AND over ranks, transform / coefficient-wise / by-N legs in lockstep, one line
    from rank 0), with the test switch that lets the two ranks share this box's GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HP_BENCH_SHARE_GPU"] = "1"
    _, r, _ = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], timeout=1800, env=env)
    assert r["n_gpus"] == 2 and r["dist_ranks"] == 2 and r["rccl_ranks"] == 0 and r["backend"] == "gloo" and "TEST MODE" in r["data"]
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == 256
    assert r["ntt"]["verified"] is True and r["coeffwise"]["mul"]["verified"] is True
    assert all(e["verified"] for e in r["ckks_by_N"].values())
    assert "cpu_baseline" not in r            # CPU legs only at N = 1
    assert r["value"] > 0 and r["config"]["batch_per_gpu"] == 256


@pytest.mark.gpu
@pytest.mark.parametrize("workload,batch", [("ckks", 256), ("bgv", 512)])
def test_eight_ranks_end_to_end_on_one_gpu(workload, batch):
    """BASELINE configs 4 and 5 as the driver launches them (`bench.py --gpus 8`): eight ranks with the per-GPU batch of the
    config (2048 / 8 = 256 CKKS pairs, 4096 / 8 = 512 BGV pairs), own seeds per rank, every rank's outputs verified, one
    line from rank 0 -- in the test mode where the ranks share this box's GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HP_BENCH_SHARE_GPU"] = "1"
    r, _, _ = run_bench(["--gpus", "8", "--workload", workload, "--steps", "2", "--warmup", "1", "--no-rates"], timeout=1800, env=env)
    assert r["n_gpus"] == 8 and r["dist_ranks"] == 8 and r["rccl_ranks"] == 0 and r["backend"] == "gloo" and "TEST MODE" in r["data"]
    assert r["scaling"] == "weak" and r["config"]["batch_per_gpu"] == batch
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == batch
    assert r["metric"] == ("ckks_hom_mult_per_s" if workload == "ckks" else "bgv_hom_mult_per_s")
    assert abs(r["value"] - 8 * batch * 1e3 / r["ms_per_step"]) < 1e-5 * r["value"]     # whole-job aggregate over the 8 ranks
    assert "cpu_baseline" not in r and "roofline" in r and "pipeline_roofline" in r
