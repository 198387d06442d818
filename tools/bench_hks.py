"""Per-stage timing of the hybrid key switch (extension).   gpurun -- 'python tools/bench_hks.py [alpha] [k] [batch]'"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import params as P
from hehub_amd.engine import Engine

alpha = int(sys.argv[1]) if len(sys.argv) > 1 else 2
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
eng = Engine(0)
logn, L = P.C3_LOGN, len(P.C3_Q)
mext = P.C3_Q + P.ntt_primes(k, logn, 50, exclude=P.C3_Q)
n, nd = 1 << logn, (L + alpha - 1) // alpha
ct1 = torch.randint(0, 1 << 39, (B, 2, L, n), dtype=torch.int64, device="cuda")
ct2 = torch.randint(0, 1 << 39, (B, 2, L, n), dtype=torch.int64, device="cuda")
key = torch.randint(0, 1 << 39, (nd, 2, L + k, n), dtype=torch.int64, device="cuda")
out = eng.empty((B, 2, L - 1, n))
f = lambda: eng.ckks_mult_hks(mext, k, alpha, ct1, ct2, key, out=out)
f(); torch.cuda.synchronize()
total = 0.0
for fam in ("tensor", "intt", "hks_modup", "ntt", "ks_inner", "hks_moddown", "hks_down_fin", "ntt_drop"):
    eng.prof_begin(fam)
    for _ in range(3):
        f()
    launches, ms = eng.prof_end()
    total += ms / 3
    print(f"{fam:13s} {launches / 3:5.1f} launches  {ms / 3:7.3f} ms per step")
print(f"sum {total:.3f} ms  (alpha={alpha}, k={k}, digits={nd}, batch={B})")
