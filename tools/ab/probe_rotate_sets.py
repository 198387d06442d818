"""A batch of rotations at the parameter sets of hehub's own benchmark (bench/benchmarks.cpp: create_params' modulus chains) through the
C ABI: ms per rotation at batch 1 / 8 / 200, to find a set whose batched path is off (examples/rotate_bench recorded: N = 4096, L = 2
was 39 us per rotation in a batch of 200 against 4.4 at N = 8192).   gpurun -- 'python tools/ab/probe_rotate_sets.py'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hehub_amd.engine import Engine

SETS = {12: (68718428161, [68714954753, 68713512961]),
        13: (17592182833153, [17592182243329, 8796090597377, 8796090007553, 8796087582721]),
        14: (2251799813554177, [2251799811391489, 281474976317441, 281474975662081, 281474974482433, 281474972188673, 281474971926529,
                                281474971533313, 281474966880257])}
eng = Engine(0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for logn, (add, q) in SETS.items():
    if only and logn != only:
        continue
    n, L = 1 << logn, len(q)
    mext = q + [add]
    key = torch.randint(0, 1 << 35, (L, 2, L + 1, n), dtype=torch.int64, device="cuda")
    for B in (1, 8, 200):
        ct = torch.randint(0, 1 << 35, (B, 2, L, n), dtype=torch.int64, device="cuda")
        for _ in range(3):
            eng.ckks_rotate(mext, ct, key, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.ckks_rotate(mext, ct, key, 1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"N={n} L={L} batch {B:3d}: {ms:8.4f} ms per call, {ms / B * 1e3:8.2f} us per rotation", flush=True)
