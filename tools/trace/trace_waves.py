"""Timeline of ALL sixteen waves of a workgroup of the C3 digit-spread launch (variant "tracew": tools/build_variant.sh tracew -DHP_TRACE
-DHP_TRACE_WAVES): per wave the stamps entry, decoded, loaded, passA, exch1, passB, exch2, passC, canon/fold, exch3, stored -- cycles since the
workgroup's first wave entered, median over the traced workgroups of the steady part of the launch; plus the SIMD each wave ran on."""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ["HEHUB_AMD_LIB"] = os.path.abspath("hehub_amd/lib_variants/libhehub_amd_tracew.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e = Engine(0)
logn, mods = 15, P.C3_MODULI_EXT
n = 1 << logn; L = len(mods) - 1
lib = capi.load()
names = "entry decoded loaded passA exch1 passB exch2 passC canon exch3 stored".split()
def dump(fn, label, lo=200, hi=1500):
    f = getattr(lib, fn); f.argtypes = [C.c_void_p, C.c_size_t]; f.restype = C.c_int
    buf = np.zeros(2048 * 16 * 12, dtype=np.uint64)
    f(buf.ctypes.data_as(C.c_void_p), buf.size)
    t = buf.reshape(2048, 16, 12).astype(np.int64)
    t = t[lo:hi]                                      # (spread launch: workgroups 3200 .. 24000 of 25600: away from the ramp and the tail)
    stamps = np.concatenate([t[:, :, 11:12], t[:, :, :10]], axis=2)   # entry first
    ok = (stamps > 0).all(axis=(1, 2))
    stamps = stamps[ok]
    hw = t[ok][:, :, 10] & 0xffffffff
    simd = (hw >> 4) & 3
    rel = stamps - stamps[:, :, 0].min(axis=1)[:, None, None]
    med = np.median(rel, axis=0)
    print(f"== {label}: {ok.sum()} workgroups; cycles since the first wave of the workgroup entered (median)")
    print("wave simd " + " ".join(f"{nm:>8s}" for nm in names))
    for w in range(16):
        sm = np.bincount(simd[:, w], minlength=4).argmax()
        print(f"{w:4d} {sm:4d} " + " ".join(f"{med[w, i]:8.0f}" for i in range(11)))
    life = (stamps[:, :, 10].max(axis=1) - stamps[:, :, 0].min(axis=1))
    print(f"workgroup lifetime median {np.median(life):.0f}  (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
    # gap between a workgroup's end and the entry of the next workgroup on the same CU is not visible here (other workgroups untraced)
B = 256
pt = torch.randint(0, 1 << 40, (B, L, n), dtype=torch.int64, device="cuda")
key = torch.randint(0, 1 << 40, (L, 2, L + 1, n), dtype=torch.int64, device="cuda")
ct = torch.randint(0, 1 << 40, (B, 2, L, n), dtype=torch.int64, device="cuda")
for lvl, fn in (("B", "hp_debug_trace"), ("A", "hp_debug_trace_a")):
    e.set_parity_level(lvl)
    for _ in range(2): e.ext_prod(mods, pt, key)
    torch.cuda.synchronize(); dump(fn, "digit-spread launch, level " + lvl)
    for _ in range(2): e.ckks_rescale(mods[:L], ct)            # 2 * 256 * 9 = 4608 items: the fused drop (rescale flavour)
    torch.cuda.synchronize(); dump(fn, "fused drop launch (rescale flavour; 'canon' = fold, 'stored' = epilogue incl. its loads), level " + lvl, 60, 240)
