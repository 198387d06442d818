"""Fraction of HBM peak of the plain transforms (in-library HIP events): forward / inverse at N = 2^logn over the C3 ciphertext moduli,
for a C3-batch-sized launch (2.5 GiB) and a steady-state launch (12.5 GiB at N = 32768).  One line: f15=.. i15=.. f15s=.. i15s=.. [f14=.. i14=..]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import params as P
from hehub_amd.engine import Engine

eng = Engine(0)
mods = P.C3_Q
L = len(mods)
out = []
for tag, logn, B in (("15", 15, 512), ("15s", 15, 2560), ("14", 14, 1024), ("13", 13, 2048)):
    n = 1 << logn
    x = torch.randint(0, 1 << 40, (B, L, n), dtype=torch.int64, device="cuda")
    for name, fam, fn in (("f", "ntt", lambda: eng.ntt_(mods, x)), ("i", "intt", lambda: eng.intt_(mods, x))):
        fn(); fn()
        eng.prof_begin(fam)
        for _ in range(10):
            fn()
        launches, ms = eng.prof_end()
        out.append(f"{name}{tag}={16.0 * n * B * L * launches / (ms * 1e-3) / 8e12:.4f}")
    del x
print(" ".join(out))
