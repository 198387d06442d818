"""Latency of one CKKS mult + relinearise + rescale call at small batches: eager launches vs a replayed HIP graph.
    gpurun -- 'python tools/bench_latency.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import params as P
from hehub_amd.engine import Engine

eng = Engine(0)
logn, mext = P.C3_LOGN, P.C3_MODULI_EXT
n, L = 1 << logn, len(mext) - 1
key = torch.randint(0, 1 << 40, (L, 2, L + 1, n), dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
print(f"C3 shape (N={n}, L={L}); ms per call")
for B in (1, 2, 4, 8, 16, 32):
    ct1 = torch.randint(0, 1 << 40, (B, 2, L, n), dtype=torch.int64, device="cuda")
    ct2 = torch.randint(0, 1 << 40, (B, 2, L, n), dtype=torch.int64, device="cuda")
    out = eng.empty((B, 2, L - 1, n))
    with torch.cuda.stream(side):
        eng.use_stream(side)
        for _ in range(3):
            eng.ckks_mult(mext, ct1, ct2, key, out=out)
        side.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.ckks_mult(mext, ct1, ct2, key, out=out)
            side.synchronize()
        eager_sync = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.ckks_mult(mext, ct1, ct2, key, out=out)
        side.synchronize()
        eager_pipe = (time.perf_counter() - t0) / reps
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            eng.ckks_mult(mext, ct1, ct2, key, out=out)
        g.replay(); side.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
            side.synchronize()
        graph_sync = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        side.synchronize()
        graph_pipe = (time.perf_counter() - t0) / reps
    print(f"batch {B:3d}: eager {eager_sync*1e3:7.3f} (sync each) {eager_pipe*1e3:7.3f} (back to back) | graph {graph_sync*1e3:7.3f} (sync each) "
          f"{graph_pipe*1e3:7.3f} (back to back) | {B/graph_pipe:8.0f} hom-mult/s")
eng.use_stream(torch.cuda.current_stream())
