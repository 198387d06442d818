"""bench.py prints ONE JSON line carrying every field of the driver's contract (small batch so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ckks", "ntt"])
def test_bench_line_has_the_contract_fields(workload):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--batch", "8", "--steps", "2",
                          "--warmup", "1", "--cpu-seconds", "1", "--cpu-procs", "0"], capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
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
        assert roof["valu_busy"] >= 0.6 and roof["traffic_frac_of_hbm_peak"] < 0.5 and roof["alu"]["valu_insts_per_wave"] > 1000
        if roof["alu"]["sclk_MHz"]:   # (a two-step region can be shorter than the 4 ms sampling period of the clock reader)
            assert 0 < roof["alu"]["frac_of_issue_peak"] < 1.2
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    cpu = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] in ("reference", "port") and cpu["cores"] == 1
    # every output of the timed buffers was compared with the CPU checker (periodic batch: 3 checker evaluations)
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == 8 and r["verify"]["checker_evaluations"] == 3
    # a one-rank run creates the RCCL communicator too (what an N-GPU job does first), and says so
    assert r["dist_ranks"] == 1 and r["rccl_ranks"] == 1 and r["rccl"]["initialised"] is True and r["rccl"]["backend"] == "nccl"


@pytest.mark.gpu
def test_default_line_carries_both_halves_of_the_metric():
    """The default command (C3, batch 256) with a short CPU budget: hom-mult/s with roofline + verification, the limb-transform
    rates at N = 4096..32768, the coefficient-wise rates, one-core and P-process CPU baselines."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1",
                          "--cpu-procs", "2", "--cpu-node-seconds", "1"], capture_output=True, text=True, timeout=1800)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
    assert REQUIRED <= set(r) and r["metric"] == "ckks_hom_mult_per_s" and r["config"]["batch_per_gpu"] == 256
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == 256
    ntt = r["ntt"]
    assert ntt["N"] == 32768 and ntt["verified"] is True and set(ntt["by_N"]) == {"4096", "8192", "16384", "32768"}
    assert ntt["steady_state"]["limbs_per_launch"] == 25600 and ntt["steady_state"]["verified"] is True
    assert ntt["steady_state"]["forward"]["frac_of_hbm_peak"] >= ntt["forward"]["frac_of_hbm_peak"] * 0.95
    for ent in ntt["by_N"].values():
        assert ent["verified"] is True
        for d in ("forward", "inverse"):
            assert ent[d]["per_s"] > 0 and 0.05 < ent[d]["frac_of_hbm_peak"] < 1 and ent[d]["achieved_GBps"] > 0
    assert set(r["ckks_by_N"]) == {"4096", "8192", "16384", "32768"}
    for ent in r["ckks_by_N"].values():
        assert ent["verified"] is True and ent["per_s"] > 0 and 0.3 < ent["A_step_frac_of_hbm_peak"] < 1
    assert r["ckks_by_N"]["4096"]["per_s"] > 4 * r["ckks_by_N"]["32768"]["per_s"]
    cw = r["coeffwise"]
    for op in ("mul", "add"):
        assert cw[op]["verified"] is True and 0.2 < cw[op]["frac_of_hbm_peak"] < 1
    node = r["cpu_baseline_node"]
    assert node["cores"] == 2 and node["value"] > r["cpu_baseline"]["value"] * 0.8 and "cores_visible" in node
    assert "cpu_quota_cores" in node and "host_limited" in node and 0 < node["parallel_efficiency"] < 1.5
    # BASELINE configs 2 and 5 at their exact shapes, and the measured stream ceiling
    c2 = r["c2"]
    assert c2["N"] == 16384 and c2["limbs"] == 4 and c2["batch_per_gpu"] == 1024 and c2["verified"] is True and c2["verified_limbs"] == 4096
    assert c2["moduli"] == [1125899904679937, 1125899903827969, 1125899903500289, 1125899903107073]
    for d in ("forward", "inverse"):
        assert c2[d]["per_s"] > 0 and 0.05 < c2[d]["roofline"]["frac"] < 1 and c2[d]["cpu_baseline"]["value"] > 0
        alu = c2[d]["roofline"].get("alu")
        if alu:   # N / 2048 = 8 waves per limb at N = 16384; nothing issues faster than the issue peak
            assert alu["waves"] == 4096 * 8 and ("frac_of_issue_peak" not in alu or 0 < alu["frac_of_issue_peak"] < 1.1)
    bgv = r["bgv"]
    assert bgv["N"] == 8192 and bgv["L"] == 6 and bgv["plain_modulus"] == 65537 and bgv["batch_per_gpu"] == 512
    assert bgv["verified"] is True and bgv["verified_outputs"] == 512 and bgv["A_step_bytes_per_op"] == 396 * 65536
    assert 0.3 < bgv["pipeline_roofline"]["frac_of_hbm_peak"] < 1 and 0.05 < bgv["roofline"]["frac"] < 1
    assert bgv["cpu_baseline"]["value"] > 0 and bgv["per_s"] > 2 * r["value"]
    assert 2000 < r["hbm_copy_ceiling_GBps"] < 8000 and r["hbm_copy"]["engine_copy_verified"] is True
    # round 4: the stream rate of every read : write mix the engine's streaming kernels have, the chip state per section, the CPU model,
    # a one-core CPU sample beside the N = 32768 transform rates, the RCCL communicator, and the opt-in parity level A in its own section
    mix = r["hbm_copy"]["stream_mix_GBps"]
    assert set(mix) == {"1R:1W", "2R:1W", "4R:3W"} and all(2000 < v < 8000 for v in mix.values())
    assert r["hbm_stream_ceiling_GBps"] >= r["hbm_copy_ceiling_GBps"]
    chip = r["chip"]
    assert {"timed_region", "ntt", "c2", "coeffwise", "bgv", "level_a"} <= set(chip)
    assert chip["timed_region"]["samples"] >= 3 and 500 < chip["timed_region"]["sclk_MHz"] <= 2500 and chip["timed_region"]["socket_power_W"] > 100
    assert isinstance(r["cpu_model"], str) and r["cpu_model"] != "unknown"
    for d in ("forward", "inverse"):
        assert ntt[d]["cpu_baseline"]["value"] > 0 and "N=32768" in ntt[d]["cpu_baseline"]["sample"]
    # round 5: the real bound of the dominant launch, every launch of a step with the measured traffic of the step, hehub's object API
    roof = r["roofline"]
    assert roof["bound"] == "valu" and roof["priced_against"] == "hbm" and 0.5 < roof["alu"]["frac_of_issue_peak"] < 1.1
    assert roof["alu"]["waves"] == 25600 * 16 and roof["alu"]["sclk_MHz"] > 500
    step = r["step"]
    assert {"tensor", "intt", "ntt", "ks_inner", "ntt_drop"} <= set(step["kernels"])
    assert abs(sum(k["ms_per_step"] for k in step["kernels"].values()) - step["kernel_ms_per_step"]) < 1e-6
    assert 0.8 * step["wall_ms_per_step"] < step["kernel_ms_per_step"] < 1.1 * step["wall_ms_per_step"]
    a_step = sum(k.get("algorithmic_bytes_per_step", 0) for k in step["kernels"].values())
    assert abs(a_step - 256 * r["config"]["A_step_bytes_per_op"]) < 1e-3 * a_step          # the families' shares add up to A_step
    tr = step["step_traffic"]
    assert 0.2 < tr["measured_over_A_step"] < 1.0 and tr["measured_over_A_min"] > 1.0 and 0.1 < tr["frac_of_hbm_peak"] < 1.0
    assert r["level_a"]["ckks"]["step"]["step_traffic"]["measured_bytes_per_op"] < tr["measured_bytes_per_op"]
    # every family against its own measured bytes: nothing moves more than the HBM peak, the families' measured bytes add up to the step's
    meas = [k["measured_bytes_per_step"] for k in step["kernels"].values() if "measured_bytes_per_step" in k]
    assert len(meas) >= 5 and abs(sum(meas) - 256 * tr["measured_bytes_per_op"]) < 0.02 * sum(meas)
    assert all(0 < k["measured_frac_of_hbm_peak"] < 1.0 for k in step["kernels"].values() if "measured_frac_of_hbm_peak" in k)
    api = r["object_api"]
    assert api["verified"] is True and api["digests_equal"] is True and api["deferred"]["digests_equal_eager"] is True
    # (rates on a shared box: loose ratios -- typically batched 27-29 k, recorded 25-27 k, single calls 12-13 k since the split-limb transforms)
    assert api["batched_call"]["per_s"] > 15000 and api["batched_call"]["per_s"] > 1.4 * api["single_calls"]["per_s"]
    assert api["deferred"]["single_calls"]["per_s"] > 1.3 * api["single_calls"]["per_s"] and api["deferred"]["fused_triples"] >= 256
    assert api["independent_chains"]["speedup"] > 1.1 and "numa_node" in r["placement"]
    mv = api["matvec"]      # the diagonal loop of matrix_vector_mul_short: every mode prints hehub's digest, recorded rotations do not lose to single calls
    assert mv["verified"] is True and set(mv["digests"]) == {"eager", "deferred", "batched-form"} and mv["ms"]["deferred"] < 1.1 * mv["ms"]["eager"]
    assert mv.get("cpu_reference_digest_equal", True) is True
    rb = api["reference_benchmark"]      # hehub's own benchmark (bench/benchmarks.cpp): hehub's digests at its four parameter sets
    assert rb["verified"] is True and set(rb["by_N"]) == {"4096", "8192", "16384", "32768"}
    assert all(e["digests_equal"] and e["look_after_every_call"] > 0 and e.get("cpu_reference_digest_equal", True) for e in rb["by_N"].values())
    la = r["level_a"]
    assert r["parity_level"] == "B"
    for k, outs in (("ckks", 256), ("bgv", 512)):
        assert la[k]["verified"] is True and la[k]["verified_outputs"] == outs and la[k]["speedup_vs_level_b"] > 1.0
    assert la["ckks"]["roofline"]["frac"] > r["roofline"]["frac"]
    for key in ("32768", "steady_32768", "c2"):     # the limb transforms as residues (hp_dev_ntt_residues / hp_dev_intt_residues)
        assert la["ntt"][key]["verified"] is True
        assert la["ntt"][key]["forward"]["frac_of_hbm_peak"] > 0.3 and la["ntt"][key]["inverse"]["frac_of_hbm_peak"] > 0.3
    assert la["ntt"]["steady_32768"]["forward"]["frac_of_hbm_peak"] > ntt["steady_state"]["forward"]["frac_of_hbm_peak"]


@pytest.mark.gpu
def test_roofline_only_line_has_no_side_legs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                          "--roofline-only"], capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
    assert "roofline" in r and not ({"ntt", "coeffwise", "cpu_baseline", "cpu_baseline_node", "verified"} & set(r))


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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=1800, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    r = json.loads(lines[0])
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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", workload, "--steps", "2",
                          "--warmup", "1", "--no-rates"], capture_output=True, text=True, timeout=1800, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["dist_ranks"] == 8 and r["rccl_ranks"] == 0 and r["backend"] == "gloo" and "TEST MODE" in r["data"]
    assert r["scaling"] == "weak" and r["config"]["batch_per_gpu"] == batch
    assert r["verified"] is True and r["verify"]["outputs_compared_per_gpu"] == batch
    assert r["metric"] == ("ckks_hom_mult_per_s" if workload == "ckks" else "bgv_hom_mult_per_s")
    assert abs(r["value"] - 8 * batch * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]     # whole-job aggregate over the 8 ranks
    assert "cpu_baseline" not in r and "roofline" in r and "pipeline_roofline" in r
