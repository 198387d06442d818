"""Phase stamps of the level-A fused drop kernels at the C3 shape (variant "trace": tools/build_variant.sh trace -DHP_TRACE):
flavour 6 (two drops in one transform: the last launch of ckks_mult) against flavour 2 (relinearize's mod-down alone)."""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ["HEHUB_AMD_LIB"] = os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e = Engine(0)
e.set_parity_level("A")
logn, mods = 15, P.C3_MODULI_EXT
n = 1 << logn; L = len(mods) - 1
lib = capi.load()
names = "decode,load,passA,exch1,passB,exch2,passC,canon,exch3,store".split(",")
def dump(W, label):
    f = lib.hp_debug_trace_a; f.argtypes = [C.c_void_p, C.c_size_t]; f.restype = C.c_int
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
    print(label, "workgroup lifetime (wave0 entry -> last wave's stores): median %.0f cycles" % np.median(full[1::2, 10] - full[0::2, 0]))
B = 256
rnd = lambda *s: torch.randint(0, 1 << 40, s, dtype=torch.int64, device="cuda")
ct1, ct2, key, quad = rnd(B, 2, L, n), rnd(B, 2, L, n), rnd(L, 2, L + 1, n), rnd(B, 3, L, n)
for _ in range(2): e.ckks_mult(mods, ct1, ct2, key)
torch.cuda.synchronize(); dump(2 * B * (L - 1), "two drops (flavour 6)")
for _ in range(2): e.ckks_relinearize(mods, quad, key)
torch.cuda.synchronize(); dump(2 * B * L, "mod-down (flavour 2) ")
