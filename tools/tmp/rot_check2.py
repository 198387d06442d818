import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import params as P
from hehub_amd.engine import Engine
eng = Engine(0)
logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
n, L = 1 << logn, len(mext) - 1
torch.manual_seed(5)
B = int(sys.argv[1])
g = lambda *s: torch.randint(0, 1 << 40, s, dtype=torch.int64, device="cuda")
ct1, ct2, key = g(B, 2, L, n), g(B, 2, L, n), g(L, 2, L + 1, n)
w = torch.arange(1, n + 1, device="cuda", dtype=torch.int64)
res = []
out = eng.ckks_mult(mext, ct1, ct2, key); res.append((out * w).sum(-1).cpu())
out.fill_(-1); eng.ckks_mult(mext, ct1, ct2, key, out=out); res.append((out * w).sum(-1).cpu())
rot = eng.ckks_rotate(mext, ct1, key, 5); res.append((rot * w).sum(-1).cpu())
rot2 = eng.ckks_rotate(mext, ct1, key, 5); res.append((rot2 * w).sum(-1).cpu())
torch.save(res, sys.argv[2])
