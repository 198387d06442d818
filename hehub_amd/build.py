"""Build the engine's shared library with hipcc for gfx950 (in-tree, no JIT cache).

    python -m hehub_amd.build            # -> hehub_amd/lib/libhehub_amd.so

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels
to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhehub_amd.so")
SOURCES = ["hp_ctx.cpp", "hp_prof.cpp", "hp_api_poly.cpp", "hp_api_scheme.cpp", "hp_api_hks.cpp", "hp_node.cpp", "hp_tables.cpp", "hp_wire.cpp",
           "hp_elem.hip", "hp_hks.hip", "hp_ntt_generic.hip", "hp_ntt_split.hip", "hp_ntt_fast.hip", "hp_ntt_a.hip"]
# the FP64 residue kernels rely on separately rounded products (error-free transformations): no contraction of a * b + c
EXTRA_FLAGS = {"hp_ntt_a.hip": ["-ffp-contract=off"]}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "hehub_amd.h"))
    objs = []
    hipcc = _hipcc()
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [spath] + headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))


HOST_LIB = os.path.join(LIBDIR, "libhehub_amd_host.so")
# the hehub-compatible host layer (hehub_amd/host/layer.hpp has the map); binding.cpp is empty in this (own-mirror) build
HOST_SOURCES = ["engine_lanes.cpp", "block_pool.cpp", "residency.cpp", "binding.cpp", "deferred_record.cpp", "deferred_run.cpp",
                "scheme_calls.cpp", "batched_forms.cpp"]


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The hehub-compatible C++ host layer (hehub_amd/host) over the C ABI; plain g++, no HIP needed."""
    build_lib(force=False, verbose=verbose)
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".hpp")] + [os.path.join(os.path.dirname(HERE), "include", "hehub_amd.h")]
    if force or _stale(HOST_LIB, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + srcs + ["-o", HOST_LIB, f"-L{LIBDIR}", "-lhehub_amd",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return HOST_LIB


def build_example(name: str, force: bool = False) -> str:
    """One of examples/*.cpp (programs written against the hehub-compatible host layer) -> examples/<name>"""
    root = os.path.dirname(HERE)
    host = build_host()
    src = os.path.join(root, "examples", name + ".cpp")
    out = os.path.join(root, "examples", name)
    deps = [src, host, os.path.join(HERE, "host", "hehub.hpp"), os.path.join(HERE, "host", "hehub_amd_ext.hpp")]
    if force or _stale(out, deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", src, "-o", out, f"-I{HERE}/host", f"-L{LIBDIR}", "-lhehub_amd_host",
                        "-lhehub_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out
