"""`bench.py --gpus N` as its own launcher, and the CPU-only self-test of the code around the timed region."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time


def self_launch(args, script: str) -> int:
    """--gpus N > 1 without a torch.distributed environment: become the launcher of N ranks on this node."""
    n = args.gpus
    if not args.launcher_selftest and not os.environ.get("HP_BENCH_SHARE_GPU"):
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} requested but only {have} HIP device(s) are visible; refusing to run on fewer",
                  file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def launcher_selftest(args) -> int:
    """The code around the timed region (rendezvous, barrier, max over ranks, one JSON line from rank 0) with gloo on CPU
    and a dummy step.  Exists so that `bench.py --gpus 2` can be exercised end to end where there is no GPU."""
    import torch.distributed as dist

    from hehub_amd import dist as hd

    world, rank, _ = hd.env_world()
    hd.init("gloo")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    step = lambda: time.sleep(0.002)
    for _ in range(args.warmup):
        step()
    hd.barrier(sync_device=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = hd.max_over_ranks(time.perf_counter() - t0)
    hd.barrier(sync_device=False)
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "value": world * args.steps / elapsed, "unit": "dummy-step/s",
                          "n_gpus": world, "dist_ranks": ranks, "rccl_ranks": 0, "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none (launcher self-test: no engine work, no claim)",
                          "config": {"workload": "launcher self-test"}}))
    hd.finalize()
    return 0
