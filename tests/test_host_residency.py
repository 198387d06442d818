"""Device-resident operands behind hehub's object API (rns.h:15-156, allocator.h:105-220).

examples/resident_chain.cpp is written against hehub's public API only: the multiply / accumulate loop of hehub's
examples/ckks_example.cpp:15-26, the rotation loop of its bench/benchmarks.cpp:31-35, rescale_inplace, then one look at the
words.  It is built three ways -- hehub alone on the CPU (oracle/_ref/ref_chain_cpu, prebuilt where the reference tree is),
hehub's headers over the binding (ref_chain_amd), and the own mirror of the interface (examples/resident_chain) -- and every
build must print the same digest of all result words.  In the own-mirror build exactly the operands and keys cross PCIe, once,
and only the polynomials the program looks at come back."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "resident_chain")
REF_CPU = os.path.join(ROOT, "oracle", "_ref", "ref_chain_cpu")
REF_AMD = os.path.join(ROOT, "oracle", "_ref", "ref_chain_amd")


def build_chain():
    from hehub_amd.build import LIBDIR, build_host

    build_host()
    src = os.path.join(ROOT, "examples", "resident_chain.cpp")
    deps = [src, os.path.join(LIBDIR, "libhehub_amd_host.so"), os.path.join(ROOT, "hehub_amd", "host", "hehub.hpp")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", src, "-o", BIN, f"-I{ROOT}/hehub_amd/host", f"-L{LIBDIR}",
                        "-lhehub_amd_host", "-lhehub_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return BIN


def run(binary, args, env=None):
    out = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    f = {"stderr": out.stderr}
    for line in out.stdout.splitlines():
        m = re.match(r"digest (\w+)", line)
        if m:
            f["digest"] = m.group(1)
        m = re.match(r"(mult\+add|rotate) per iteration ([\d.]+) ms", line)
        if m:
            f[m.group(1)] = float(m.group(2))
        m = re.match(r"pcie (to_device|to_host) ([\d.]+) MiB in (\d+) copies \(.* = ([\d.]+) MiB\)", line)
        if m:
            f[m.group(1)] = (float(m.group(2)), int(m.group(3)), float(m.group(4)))
    return f


def test_chain_program_builds():
    assert os.path.exists(build_chain())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(15, 10, 24), (12, 4, 5), (13, 2, 3), (11, 7, 2)])
def test_resident_chain_matches_hehub_and_crosses_pcie_once(shape):
    own = run(build_chain(), shape)
    assert "digest" in own
    # exactly the operands and the two keys went up, exactly the polynomials that were looked at came back
    assert own["to_device"][0] == own["to_device"][2], own
    assert own["to_host"][0] == own["to_host"][2], own
    if os.path.exists(REF_CPU):          # hehub itself, on the CPU: the words every build must reproduce
        assert run(REF_CPU, shape)["digest"] == own["digest"]
    if os.path.exists(REF_AMD):          # hehub's own headers over the binding, with and without the opt-in caches
        for env in ({}, {"HEHUB_AMD_CT_CACHE": "64", "HEHUB_AMD_KEY_CACHE": "4"}):
            assert run(REF_AMD, shape, env)["digest"] == own["digest"], env
    # the same accounting and the same words when every call runs as it is made
    eager = run(build_chain(), shape, {"HEHUB_AMD_DEFER": "0"})
    assert eager["digest"] == own["digest"] and eager["to_device"][0] == eager["to_device"][2] and eager["to_host"][0] == eager["to_host"][2], eager


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_AMD), reason="oracle/_ref/ref_chain_amd is only built where the reference tree exists")
def test_binding_ct_cache_saves_the_uploads():
    """hehub's own types are host memory, so results always come back; with HEHUB_AMD_CT_CACHE an operand that an earlier call
    uploaded or produced is not uploaded again: the bytes to the device drop to (about) operands + keys once"""
    shape = (13, 4, 6)
    n, L = 1 << shape[0], shape[1]
    def mib(env):
        r = run(REF_AMD, shape, dict(env, HEHUB_AMD_VERBOSE="1"))
        m = re.search(r"PCIe: (\d+) copies / ([\d.]+) MiB to the device, (\d+) copies / ([\d.]+) MiB back", r["stderr"])
        assert m, r["stderr"]
        return float(m.group(2)), float(m.group(4)), r["digest"]
    up0, down0, d0 = mib({})
    up1, down1, d1 = mib({"HEHUB_AMD_CT_CACHE": "64", "HEHUB_AMD_KEY_CACHE": "4"})
    once = (2 * 2 * L + 2 * (2 * L * (L + 1))) * n * 8 / 1048576.0
    # hehub's own ckks::add is `auto r(a); r += b` (rlwe.cpp:83-85 -> rns.h:218-222): the copy r is a NEW host object the cache has never seen,
    # so each accumulation uploads one ciphertext; everything else (operands, keys, products, rotations) goes up once
    copies = (shape[2] - 1) * 2 * L * n * 8 / 1048576.0
    assert d0 == d1 and abs(down0 - down1) < 0.2
    assert up1 <= once + copies + 0.1 and up0 > 3 * up1, (up0, up1, once, copies)
