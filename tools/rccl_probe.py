#!/usr/bin/env python3
"""RCCL all_gather bandwidth between the GPUs of one node (tools/first_node_run.sh step 6): per message size the time of
dist.all_gather_into_tensor and the bus bandwidth (bytes received per rank / time).  One rank: the communicator still comes up."""
import os
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29672")
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
for lg in range(20, 29, 2):
    n = (1 << lg) // 8
    src = torch.full((n,), rank, dtype=torch.int64, device="cuda")
    dst = torch.empty((world * n,), dtype=torch.int64, device="cuda")
    for _ in range(3):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = bool((dst.view(world, n)[:, 0].cpu() == torch.arange(world)).all())
    if rank == 0:
        print(f"all_gather {1 << lg:>10} B per rank x {world} ranks: {dt * 1e6:8.1f} us, bus {((world - 1) * (1 << lg)) / dt / 1e9:7.2f} GB/s per rank, correct {ok}")
dist.destroy_process_group()
