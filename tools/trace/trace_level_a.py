"""Phase stamps of k_ntt_fwd_a<15> in the C3 digit-spread launch (variant "trace": tools/build_variant.sh trace -DHP_TRACE)."""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ["HEHUB_AMD_LIB"] = os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e = Engine(0)
logn, mods = 15, P.C3_MODULI_EXT
n = 1 << logn; L = len(mods) - 1
lib = capi.load()
names = "decode,load,passA,exch1,passB,exch2,passC,canon,exch3,store".split(",")
def dump(fn, W, label):
    f = getattr(lib, fn); f.argtypes = [C.c_void_p, C.c_size_t]; f.restype = C.c_int
    nrec = ((W + 15) // 16) * 2
    buf = np.zeros(4096 * 12, dtype=np.uint64)
    f(buf.ctypes.data_as(C.c_void_p), buf.size)
    t = buf.reshape(-1, 12)[:min(nrec, 4096)].astype(np.int64)
    full = np.concatenate([t[:, 11:12], t[:, :10]], axis=1)     # entry stamp first
    d = np.diff(full, axis=1)
    for w, lab in ((0, "wave0"), (1, "lastwave")):
        dd = d[w::2]
        print(label, f"W={W:5d}", lab, " ".join(f"{nm}={np.median(dd[:, i]):.0f}" for i, nm in enumerate(names)),
              "total=%.0f" % np.median(full[w::2, 10] - full[w::2, 0]))
    # lifetime of a workgroup: entry of wave 0 to the last stamp of the last wave
    print(label, "workgroup lifetime (wave0 entry -> last wave's stores): median %.0f cycles" % np.median(full[1::2, 10] - full[0::2, 0]))
B = 256
pt = torch.randint(0, 1 << 40, (B, L, n), dtype=torch.int64, device="cuda")
key = torch.randint(0, 1 << 40, (L, 2, L + 1, n), dtype=torch.int64, device="cuda")
for lvl, fn in (("B", "hp_debug_trace"), ("A", "hp_debug_trace_a")):
    e.set_parity_level(lvl)
    for _ in range(2): e.ext_prod(mods, pt, key)
    torch.cuda.synchronize(); dump(fn, B * (L * L), "spread level " + lvl)
