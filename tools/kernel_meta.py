"""Resource usage of every kernel in the built engine library, read from the code objects' metadata (no recompilation):
    python tools/kernel_meta.py [path/to/libhehub_amd.so] [name filter]
prints  name  vgpr  vgpr_spill  sgpr_spill  scratch_bytes  lds_bytes.  Used by tests/test_kernel_resources.py."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(lib, tmp):
    """the gfx950 code objects bundled in the library (one per translation unit), as files under tmp"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for n, s in enumerate(starts):
        blob = os.path.join(tmp, f"b{n}.bin")
        open(blob, "wb").write(data[s:starts[n + 1] if n + 1 < len(starts) else len(data)])
        co = os.path.join(tmp, f"co{n}.o")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={blob}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        out.append(co)
    return out


def disassembly(lib=None):
    """llvm-objdump -d of every code object, concatenated"""
    lib = lib or os.path.join(ROOT, "hehub_amd", "lib", "libhehub_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        return "\n".join(subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
                         for co in code_objects(lib, tmp))


def kernel_meta(lib=None):
    lib = lib or os.path.join(ROOT, "hehub_amd", "lib", "libhehub_amd.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
        for n, s in enumerate(starts):
            blob = os.path.join(tmp, f"b{n}.bin")
            open(blob, "wb").write(data[s:starts[n + 1] if n + 1 < len(starts) else len(data)])
            co = os.path.join(tmp, f"co{n}.o")
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={blob}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count" and cur.get("name"):   # first key of the next kernel's record
                    cur = {}
                if k in ("name", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                         "group_segment_fixed_size", "sgpr_count"):
                    cur[k] = v if k == "name" else int(v)
                if k == "name":
                    out[v] = cur
    names = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True).stdout.splitlines()
    return {d: out[m] for m, d in zip(list(out), names)}


if __name__ == "__main__":
    meta = kernel_meta(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None)
    flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    for name, r in sorted(meta.items()):
        if flt in name:
            print(f"{name[:90]:90s} vgpr={r.get('vgpr_count')} spill={r.get('vgpr_spill_count')} sgpr_spill={r.get('sgpr_spill_count')} "
                  f"scratch={r.get('private_segment_fixed_size')} lds={r.get('group_segment_fixed_size')}")
