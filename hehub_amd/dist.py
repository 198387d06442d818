"""Multi-GPU plumbing for the batch-sharded mode.

Ciphertexts are independent units: a node-level batch is cut into contiguous slices, one per rank
(one process per GPU), with NO collective on the data path -- keys and twiddle tables are read-only
and replicated.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests) is only
used to rendezvous, to fence the timed region and to take the max elapsed time over ranks.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class _StdoutToStderr:
    """RCCL prints a version banner on fd 1 when its first communicator is created; a bench line is ONE JSON line on stdout.
    While the group is created (and its first collective runs) fd 1 points at fd 2."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import sys

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)   # the banner is a C printf: it sits in libc's buffer when fd 1 is a pipe
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init(backend: str, device=None, force: bool = False):
    """Rendezvous.  A one-rank job needs no process group; `force` (bench.py --force-dist, HP_FORCE_DIST=1) creates it anyway,
    so that a one-GPU box runs exactly what an N-GPU job runs: the RCCL communicator, device-side barrier / all_reduce."""
    import torch.distributed as dist

    world, rank, _ = env_world()
    force = force or bool(os.environ.get("HP_FORCE_DIST"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket

            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL peer buffers)
        kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
        with _StdoutToStderr():
            dist.init_process_group(backend, **kwargs)
            if backend == "nccl":
                dist.barrier()     # the first collective creates the communicator (and prints RCCL's banner)
    return world, rank


def barrier(sync_device: bool = True):
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync_device and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def finalize():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
