#!/bin/bash
# interleaved A/B of the inverse (and forward) transforms at N = 2048 .. 16384: tools/ab/ab_small_inv.sh <reps> <variant> ...  ("main" = hehub_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
REPS=$1; shift
for i in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset HEHUB_AMD_LIB; else export HEHUB_AMD_LIB=$R/hehub_amd/lib_variants/libhehub_amd_$v.so; fi
    echo "$v $(python - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import torch, params as P
from hehub_amd.engine import Engine
eng = Engine(0); mods = P.C3_Q; L = len(mods); out = []
for logn in (11, 12, 13, 14):
    n = 1 << logn; B = (1 << 27) // (L * n)
    x = torch.randint(0, 1 << 40, (B, L, n), dtype=torch.int64, device="cuda")
    for name, fam, fn in (("i", "intt", lambda: eng.intt_(mods, x)), ("is", "intt", lambda: eng.intt_(mods, x, strict=True))):
        fn(); fn(); eng.prof_begin(fam)
        for _ in range(10): fn()
        launches, ms = eng.prof_end()
        out.append(f"{name}{logn}={16.0 * n * B * L * launches / (ms * 1e-3) / 8e12:.4f}")
print(" ".join(out))
PY
)"
  done
done | python3 -c "
import sys, collections, statistics
d=collections.defaultdict(lambda: collections.defaultdict(list))
for line in sys.stdin:
    t=line.split()
    for kv in t[1:]:
        k,v=kv.split('='); d[t[0]][k].append(float(v))
for v,fam in d.items():
    print(v, ' '.join(f'{k}={statistics.median(x):.4g}' for k,x in fam.items()), f'(n={len(next(iter(fam.values())))})')
"
