"""Throughput through hehub's OBJECT API (ckks.h:270-313 is one ciphertext per call, its callers loop over
independent ciphertexts: src/circuits/linear_algebra.h:109-133, bench/benchmarks.cpp:24-35).

examples/independent_mults.cpp runs B independent ckks::mult + rescale_inplace three ways -- the loop of single calls, the batched
form of hehub_amd_ext.hpp (one engine call), independent chains of single calls over the layer's lanes -- and prints a digest of
every result word per mode.  All modes, the build against hehub's own headers over the binding, and hehub itself on the CPU
(oracle/_ref/ref_indep_cpu, prebuilt where the reference tree is) must print the same digests."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "independent_mults")
REF_CPU = os.path.join(ROOT, "oracle", "_ref", "ref_indep_cpu")
REF_AMD = os.path.join(ROOT, "oracle", "_ref", "ref_indep_amd")


EAGER = {"HEHUB_AMD_DEFER": "0"}      # every call runs when it is made (the layer's default since round 6 is to record)
RECORDED = {"HEHUB_AMD_DEFER": "1"}   # == the default; named so that a test says what it runs


def build_example():
    from hehub_amd.build import LIBDIR, build_host

    build_host()
    src = os.path.join(ROOT, "examples", "independent_mults.cpp")
    deps = [src, os.path.join(LIBDIR, "libhehub_amd_host.so"), os.path.join(ROOT, "hehub_amd", "host", "hehub.hpp"),
            os.path.join(ROOT, "hehub_amd", "host", "hehub_amd_ext.hpp")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", src, "-o", BIN, f"-I{ROOT}/hehub_amd/host", f"-L{LIBDIR}",
                        "-lhehub_amd_host", "-lhehub_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return BIN


def run(binary, args, env=None, drop=()):
    base = {k: v for k, v in os.environ.items() if k not in drop}
    out = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, timeout=1800,
                         env=dict(base, **(env or {})))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    f = {}
    for line in out.stdout.splitlines():
        m = re.match(r"([\w-]+) digest (\w+)", line)
        if m:
            f[m.group(1)] = m.group(2)
        m = re.match(r"(serial|batch) ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\)", line)
        if m:
            f[m.group(1) + "_per_s"] = float(m.group(3))
        m = re.match(r"chains .* ([\d.]+) ms per step on 1 lane, ([\d.]+) ms on (\d+) lanes", line)
        if m:
            f["chain_ms"] = (float(m.group(1)), float(m.group(2)), int(m.group(3)))
        m = re.match(r"deferred: (\d+) recorded calls ran as (\d+) batched engine calls \((\d+) ", line)
        if m:
            f["recorded"], f["groups"], f["fused"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
    return f


def test_example_builds():
    assert os.path.exists(build_example())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 4, 6), (13, 6, 9), (11, 2, 3)])
def test_every_mode_prints_hehubs_words(shape):
    args = list(shape) + ["all", 2, 8, 3, 2]
    own = run(build_example(), args, EAGER)
    assert own["serial"] == own["batch"], own                  # B single calls == one batched call
    assert own["serial-chain"] == own["batch-chain"], own      # rotate + add + rescale, single calls == batched
    assert own["chains"] == own["chains-lanes"], own           # independent chains: 1 lane == 8 lanes
    if os.path.exists(REF_CPU):                                # hehub itself on the CPU
        ref = run(REF_CPU, list(shape) + ["all", 1, 8, 3, 2])
        for k in ("serial", "serial-chain", "chains"):
            assert ref[k] == own[k], (k, ref, own)
    if os.path.exists(REF_AMD):                                # hehub's own headers over the binding + hehub_amd_ext.hpp
        bind = run(REF_AMD, args)
        for k in ("serial", "batch", "serial-chain", "batch-chain", "chains"):
            assert bind[k] == own[k], (k, bind, own)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 4, 6), (13, 6, 9)])
def test_deferred_mode_prints_the_same_words(shape):
    """the layer's default: the scheme-level calls are recorded and run as batches (hehub_amd/host/deferred_record.cpp, deferred_run.cpp) -- every mode of the
    program must print what the call-by-call run (HEHUB_AMD_DEFER=0) prints, with and without the variable in the environment"""
    args = list(shape) + ["all", 2, 8, 3, 2]
    eager = run(build_example(), args, EAGER)
    assert "recorded" not in eager
    lazy = run(build_example(), args, RECORDED)
    default = run(build_example(), args, {}, drop=("HEHUB_AMD_DEFER",))
    assert default.get("recorded", 0) > 0 and lazy.get("recorded", 0) > 0
    for k in ("serial", "batch", "serial-chain", "batch-chain", "chains", "chains-lanes"):
        assert default[k] == eager[k], (k, default, eager)
    for k in ("serial", "batch", "serial-chain", "batch-chain", "chains", "chains-lanes"):
        assert lazy[k] == eager[k], (k, lazy, eager)


@pytest.mark.gpu
def test_modes_agree_at_parity_level_a():
    """HP_PARITY_LEVEL=A (canonical residues): single calls, the batched form (fused pipeline: two drops as one transform), lanes and
    deferred mode (recorded triples run as the fused pipeline) all return the same words -- residues have one representative"""
    args = [13, 6, 9, "all", 2, 8, 3, 2]
    eager = run(build_example(), args, {"HP_PARITY_LEVEL": "A", "HEHUB_AMD_DEFER": "0"})
    lazy = run(build_example(), args, {"HP_PARITY_LEVEL": "A", "HEHUB_AMD_DEFER": "1"})
    level_b = run(build_example(), args, EAGER)
    assert eager["serial"] == eager["batch"] and eager["serial-chain"] == eager["batch-chain"] and eager["chains"] == eager["chains-lanes"]
    for k in ("serial", "batch", "serial-chain", "batch-chain", "chains"):
        assert lazy[k] == eager[k], (k, lazy, eager)
    assert eager["serial"] != level_b["serial"]          # (some lazy word of hehub's is >= q: level A is not level B)


@pytest.mark.gpu
def test_c3_unchanged_loop_in_deferred_mode():
    """the loop of single calls (hehub's interface as it is) in deferred mode: the recorded mult + rescale_inplace pairs run as the
    engine's fused batch pipeline -- the batch rate without a source change (24 - 28 k hom-mult/s at B = 256; loose bound here)"""
    eager = run(build_example(), [15, 10, 64, "serial", 3], EAGER)
    lazy = run(build_example(), [15, 10, 64, "serial", 3], {}, drop=("HEHUB_AMD_DEFER",))   # nothing in the environment
    assert lazy["serial"] == eager["serial"] and lazy["serial-chain"] == eager["serial-chain"]
    assert lazy["fused"] >= 64, lazy     # (the rates: tests/test_perf.py)


@pytest.mark.gpu
def test_c3_batched_form_reaches_the_engine_rate():
    """C3 (N = 32768, L = 10) through hehub's types: the batched form must deliver the engine's batch rate (29 k hom-mult/s on an
    MI355X at B = 256, 24 k at the B = 64 used here; the bounds are loose: shared boxes), the loop of single calls is latency-bound
    (3 k on one lane, 6 k over the default four), and lanes make independent chains overlap"""
    r = run(build_example(), [15, 10, 64, "all", 3, 8, 8, 4], EAGER)
    assert r["serial"] == r["batch"] and r["chains"] == r["chains-lanes"], r     # (the rates: tests/test_perf.py)
