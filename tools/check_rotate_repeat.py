"""Repeat C3 batch-256 operations (rotation, ckks mult, and a C5-shape bgv mult) and compare every run with the first one word for word
(GPU box): a non-deterministic kernel shows up as rows that differ between runs.  python tools/check_rotate_repeat.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import params as P
from hehub_amd.engine import Engine
eng = Engine(0)
logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
n, L = 1 << logn, len(mext) - 1
torch.manual_seed(5)
B = 256
g = lambda *s: torch.randint(0, 1 << 40, s, dtype=torch.int64, device="cuda")
ct1, ct2, key = g(B, 2, L, n), g(B, 2, L, n), g(L, 2, L + 1, n)
out = eng.ckks_mult(mext, ct1, ct2, key)
ref = eng.ckks_rotate(mext, ct1, key, 5).clone()
for it in range(12):
    r = eng.ckks_rotate(mext, ct1, key, 5)
    bad = (r != ref)
    rows = bad.any(-1).nonzero()
    print("iter", it, "bad rows", rows.tolist()[:6])
    for (c, p, k) in rows.tolist()[:3]:
        idx = bad[c, p, k].nonzero().flatten()
        print("   row", c, p, k, "n bad", len(idx), "first", idx[:6].tolist(), "last", idx[-3:].tolist(), "spacing", (idx[1:] - idx[:-1]).unique().tolist()[:6])

ref = eng.ckks_mult(mext, ct1, ct2, key).clone()
for it in range(12):
    r = eng.ckks_mult(mext, ct1, ct2, key)
    print("mult iter", it, "bad rows", (r != ref).any(-1).nonzero().tolist()[:6])
del ref, r, ct1, ct2, key, out
logn, mext = P.C5_LOGN, P.C5_MODULI_EXT
n, L, B = 1 << logn, len(mext) - 1, 512
b1, b2, bk = g(B, 2, L, n), g(B, 2, L, n), g(L, 2, L + 1, n)
ref = eng.bgv_mult(mext, P.C5_T, b1, b2, bk).clone()
for it in range(12):
    r = eng.bgv_mult(mext, P.C5_T, b1, b2, bk)
    print("bgv iter", it, "bad rows", (r != ref).any(-1).nonzero().tolist()[:6])
