"""PCIe-inclusive rate of the node layer's host-resident entry point (hp_node_ckks_mult_relin_rescale): numpy batches in host
memory in, numpy out -- what a hehub-side C++ host with host-resident ciphertexts gets.   python tools/bench_node.py [ranks] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import params as P
from hehub_amd.node import Node

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mext, logn = P.C3_MODULI_EXT, P.C3_LOGN
n, L = 1 << logn, len(mext) - 1
rs = np.random.RandomState(1)
ct1 = rs.randint(0, 1 << 39, (B, 2, L, n)).astype(np.uint64)
ct2 = rs.randint(0, 1 << 39, (B, 2, L, n)).astype(np.uint64)
key = rs.randint(0, 1 << 39, (L, 2, L + 1, n)).astype(np.uint64)
node = Node([0] * ranks)
dk = node.replicate(key)
gb = (ct1.nbytes * 2 + B * 2 * (L - 1) * n * 8) / 1e9
for kind in ("pageable", "page-locked (hp_host_alloc)"):
    if kind != "pageable":
        p1, p2, po = node.pinned(ct1.shape), node.pinned(ct2.shape), node.pinned((B, 2, L - 1, n))
        p1[...] = ct1; p2[...] = ct2
        a, b, o = p1, p2, po
    else:
        a, b, o = ct1, ct2, np.empty((B, 2, L - 1, n), dtype=np.uint64)
    ref = node.ckks_mult(mext, a, b, dk, out=o).copy()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        node.ckks_mult(mext, a, b, dk, out=o)
    dt = (time.perf_counter() - t0) / reps
    print(f"node host path: ranks={ranks} (sharing GPU 0) batch={B}, {kind} host memory: {B/dt:.0f} hom-mult/s, {gb/dt:.1f} GB/s over PCIe, {dt*1e3:.1f} ms per call")
    if kind == "pageable":
        first = ref
    else:
        assert np.array_equal(first, ref)
node.close()
