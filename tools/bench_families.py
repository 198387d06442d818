"""Per-kernel-family time of one C3 hom-mult step (in-library HIP events, one pass per family) plus the whole-step rate.
    [HEHUB_AMD_LIB=<variant>] python tools/bench_families.py [--workload ckks|bgv] [--batch B] [--reps R]
prints one line: family=ms ... step=ms rate=hom-mult/s"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import params as P
from hehub_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ckks"); ap.add_argument("--logn", type=int, default=0); ap.add_argument("--batch", type=int, default=0); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
eng = Engine(0)
if a.workload == "ckks":
    logn, mext, B = a.logn or P.C3_LOGN, P.C3_MODULI_EXT, a.batch or 256
    run = lambda: eng.ckks_mult(mext, ct1, ct2, key, out=out)
else:
    logn, mext, B = P.C5_LOGN, P.C5_MODULI_EXT, a.batch or 512
    run = lambda: eng.bgv_mult(mext, P.C5_T, ct1, ct2, key, out=out)
n, L = 1 << logn, len(mext) - 1
g = lambda *s: torch.randint(0, 1 << 40, s, dtype=torch.int64, device="cuda")
ct1, ct2, key, out = g(B, 2, L, n), g(B, 2, L, n), g(L, 2, L + 1, n), eng.empty((B, 2, L - 1, n))
run(); run(); torch.cuda.synchronize()
res = {}
for fam in ("tensor", "intt", "ntt", "ks_inner", "ntt_drop"):
    eng.prof_begin(fam)
    for _ in range(a.reps):
        run()
    launches, ms = eng.prof_end()
    res[fam] = ms / a.reps
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.reps * 2):
    run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (a.reps * 2)
print(" ".join(f"{k}={v:.3f}" for k, v in res.items()), f"step={dt*1e3:.3f} rate={B/dt:.0f}")
