"""The limb-sharded plan of the C node layer (hp_node_sharded_mult_dev) under its three transports -- direct peer writes, RCCL
(ncclAllGather / ncclBroadcast, librccl loaded on demand), the packed buffers moved by copies: ms per C3 multiplication and the words
compared with the peer transport's.   python tools/node_transports.py [ranks = devices visible] [batch = 8] [share = 0]
share = 1: the ranks share GPU 0 (RCCL is then refused, as it must be)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import params as P
from hehub_amd.engine import HpError
from hehub_amd.node import Node, ShardedPlan

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else torch.cuda.device_count()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
share = len(sys.argv) > 3 and sys.argv[3] == "1"
mext, logn = P.C3_MODULI_EXT, P.C3_LOGN
n, L = 1 << logn, len(mext) - 1
rs = np.random.RandomState(7)
ct1 = rs.randint(0, 1 << 39, (B, 2, L, n)).astype(np.uint64)
ct2 = rs.randint(0, 1 << 39, (B, 2, L, n)).astype(np.uint64)
key = rs.randint(0, 1 << 39, (L, 2, L + 1, n)).astype(np.uint64)
devs = [0] * ranks if share else list(range(ranks))
node = Node(devs)
dk = node.replicate(key)
tdev = lambda a, d: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(f"cuda:{d}")
r1 = [tdev(ct1, d) for d in devs]; r2 = [tdev(ct2, d) for d in devs]
first = None
for name in ("peer", "packed", "rccl"):
    try:
        node.set_transport(name)
    except HpError as e:
        print(f"transport {name}: refused ({e.msg[:120]})")
        continue
    plan = ShardedPlan(node, logn, mext, B)
    ro = [torch.zeros((B, 2, L - 1, n), dtype=torch.int64, device=f"cuda:{d}") for d in devs]
    torch.cuda.synchronize()
    plan.mult_dev(r1, r2, dk, ro)          # warm: tables, workspaces, the communicator's first collective
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps):
        plan.mult_dev(r1, r2, dk, ro)
    dt = (time.perf_counter() - t0) / reps
    got = ro[0].cpu().numpy().view(np.uint64)
    same = all(np.array_equal(x.cpu().numpy().view(np.uint64), got) for x in ro)
    if first is None:
        first = got
    print(f"transport {name}: ranks={ranks}{' (sharing GPU 0)' if share else ''} batch={B}: {dt * 1e3:.2f} ms per multiplication of the batch "
          f"({B / dt:.0f} hom-mult/s); every rank holds the same result: {same}; words equal to the peer transport's: {np.array_equal(got, first)}")
    plan.close()
node.close()
