"""Per-stage timing through the C ABI (in-library HIP events): compares transform launch shapes.
    gpurun -- 'python tools/bench_stage.py'"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import params as P
from hehub_amd.engine import Engine

eng = Engine(0)
logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
n, L = 1 << logn, len(mext) - 1


def timed(family, fn, reps=5):
    fn(); torch.cuda.synchronize()
    eng.prof_begin(family)
    for _ in range(reps):
        fn()
    launches, ms = eng.prof_end()
    return launches / reps, ms / reps


B = int(os.environ.get("B", "256"))
pt = torch.randint(0, 1 << 40, (B, L, n), dtype=torch.int64, device="cuda")
key = torch.randint(0, 1 << 40, (L, 2, L + 1, n), dtype=torch.int64, device="cuda")
l, ms = timed("ntt", lambda: eng.ext_prod(mext, pt, key))
limbs = B * L * L
print(f"spread NTT   : {limbs} limbs  {ms:.3f} ms  {ms*1e3/limbs:.4f} us/limb  {limbs*16*n/ms/1e6:.0f} GB/s  ({l} launches)")
l, ms = timed("ks_inner", lambda: eng.ext_prod(mext, pt, key))
print(f"ks_inner     : {ms:.3f} ms  {(B*(L+1)*(L+2)*8*n)/ms/1e6:.0f} GB/s (digits+out)")
Bb = limbs // (L + 1)
x = torch.randint(0, 1 << 40, (Bb, L + 1, n), dtype=torch.int64, device="cuda")
l, ms = timed("ntt", lambda: eng.ntt_(mext, x))
lb = Bb * (L + 1)
print(f"in-place NTT : {lb} limbs  {ms:.3f} ms  {ms*1e3/lb:.4f} us/limb  {lb*16*n/ms/1e6:.0f} GB/s")
