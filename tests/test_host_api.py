"""The C++ host layer (hehub_amd/host/hehub.hpp): hehub's own unit tests for this path restated in
tests/cpp/host_api_test.cpp.  Without a GPU the test only proves that the mirrored interface
compiles and links against the C ABI; with one (-m gpu) it runs the binary."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_api_test")


def build_binary():
    from hehub_amd.build import LIBDIR, build_host
    from oracle.pyoracle import build as build_oracle

    build_host()
    build_oracle(ref=False)
    src = os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp")
    deps = [src, os.path.join(LIBDIR, "libhehub_amd_host.so"), os.path.join(ROOT, "hehub_amd", "host", "hehub.hpp")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", src, "-o", BIN, f"-I{ROOT}/hehub_amd/host", f"-L{LIBDIR}",
                        "-lhehub_amd_host", "-lhehub_amd", f"-L{ROOT}/oracle", "-lhehub_oracle",
                        f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ROOT}/oracle", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return BIN


def test_host_layer_compiles_and_links():
    assert os.path.exists(build_binary())


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"HEHUB_AMD_DEFER": "0"}, {"HEHUB_AMD_DEFER": "0", "HEHUB_AMD_LANES": "1"}],
                         ids=["default-recorded", "call-by-call", "call-by-call-one-lane"])
def test_host_layer_runs_reference_unit_tests(env):
    """hehub's unit tests for this path over the mirror, in the layer's default mode (calls recorded and run grouped, since round 6) and
    call by call"""
    base = {k: v for k, v in os.environ.items() if k not in ("HEHUB_AMD_DEFER", "HEHUB_AMD_LANES")}
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=600, env=dict(base, **env))
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "All tests passed" in out.stdout


REF_TESTS = os.path.join(ROOT, "oracle", "_ref", "ref_tests_amd")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_TESTS), reason="oracle/_ref/ref_tests_amd is only built where the reference tree exists (make -C oracle ref_tests)")
def test_reference_test_suite_over_the_binding():
    """hehub's own Catch2 suite (tests/*.cpp, 481 assertions), linked with hehub's own sampling / encoding /
    key generation and with hehub_amd/host/*.cpp (-DHEHUB_AMD_BIND_REFERENCE) in place of hehub's hot-path
    definitions.  The binary is prebuilt by oracle/Makefile; nothing is read from the reference tree here."""
    env = dict(os.environ, HEHUB_AMD_VERBOSE="1")
    out = subprocess.run([REF_TESTS], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "All tests passed" in out.stdout
    calls = [int(l.split()[1]) for l in out.stderr.splitlines() if l.startswith("hehub_amd:") and "engine calls" in l]
    assert calls and calls[0] > 1000, "the suite must actually have gone through the device engine"


REF_E2E = os.path.join(ROOT, "oracle", "_ref", "ref_e2e_amd")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_E2E), reason="oracle/_ref/ref_e2e_amd is only built where the reference tree exists (make -C oracle ref_e2e)")
def test_end_to_end_tolerance_at_baseline_shapes():
    """tests/cpp/ref_e2e.cpp: hehub's own keygen / encode / encrypt / decrypt around the device hot path.  CKKS C3 shape
    (N=32768, L=10): mult + relinearize + rescale max slot error <= 2^-24; BGV C5 shape: plaintext-level checks."""
    out = subprocess.run([REF_E2E], capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "All end-to-end checks passed" in out.stdout
    # with the extensions on, the same relinearisation key serves a second multiplication one level down
    out = subprocess.run([REF_E2E], capture_output=True, text=True, timeout=900, env=dict(os.environ, HEHUB_AMD_EXTENSIONS="1"))
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0 and "ok    second multiplication with the same key" in out.stdout


EXAMPLE_SRC = os.path.join(ROOT, "examples", "ckks_throughput.c")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "ckks_throughput")


def build_example():
    from hehub_amd.build import LIBDIR, build_lib

    build_lib(verbose=False)
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-Werror", EXAMPLE_SRC, f"-I{ROOT}/include", f"-L{LIBDIR}",
                    "-lhehub_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXAMPLE_BIN], check=True)
    return EXAMPLE_BIN


def test_plain_c_example_builds():
    """examples/ckks_throughput.c uses nothing but include/hehub_amd.h (C99): the C ABI is self-sufficient."""
    assert os.path.exists(build_example())


@pytest.mark.gpu
def test_plain_c_example_runs():
    out = subprocess.run([build_example(), "12", "8", "3"], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "hom-mult/s" in out.stdout


MULTI_SRC = os.path.join(ROOT, "examples", "multi_ctx.c")
MULTI_BIN = os.path.join(ROOT, "examples", "multi_ctx")


def build_multi():
    from hehub_amd.build import LIBDIR, build_lib

    build_lib(verbose=False)
    subprocess.run(["gcc", "-O2", "-std=c99", "-pthread", "-Wall", "-Wextra", "-Werror", MULTI_SRC, f"-I{ROOT}/include",
                    f"-L{LIBDIR}", "-lhehub_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", MULTI_BIN], check=True)
    return MULTI_BIN


def test_multi_context_example_builds():
    assert os.path.exists(build_multi())


@pytest.mark.gpu
def test_contexts_are_independent_across_threads():
    """examples/multi_ctx.c: four host threads, four engine contexts sharing GPU 0, identical results everywhere."""
    out = subprocess.run([build_multi(), "4", "1", "12", "3", "2"], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "digest" in out.stdout and "everywhere" in out.stdout


NODE_SRC = os.path.join(ROOT, "examples", "node_batch.c")
NODE_BIN = os.path.join(ROOT, "examples", "node_batch")


def build_node_example():
    from hehub_amd.build import LIBDIR, build_lib

    build_lib(verbose=False)
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-Werror", NODE_SRC, f"-I{ROOT}/include", f"-L{LIBDIR}", "-lhehub_amd",
                    f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", NODE_BIN], check=True)
    return NODE_BIN


def test_node_example_builds():
    assert os.path.exists(build_node_example())


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 5])
def test_node_example_drives_several_contexts_from_c(ranks):
    """examples/node_batch.c (C99, the C ABI alone): a host batch through the node API -- batch-sharded and limb-sharded over
    `ranks` contexts -- equals the single-context result word for word"""
    out = subprocess.run([build_node_example(), str(ranks), "1", "12", "6"], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "single context: yes" in out.stdout
