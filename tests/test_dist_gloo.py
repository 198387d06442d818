"""The N > 1 path on CPU: two processes, gloo backend (world_size 2), exercising exactly the helpers
bench.py uses around its timed region (rendezvous, barrier, max-over-ranks) and the batch sharding."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from hehub_amd import dist as hd
world, rank = hd.init("gloo")
assert world == 2
lo, hi = hd.shard_range(2048, world, rank)
assert (lo, hi) == ((0, 1024) if rank == 0 else (1024, 2048))
hd.barrier(sync_device=False)
elapsed = 0.25 if rank == 0 else 0.75
m = hd.max_over_ranks(elapsed)
assert abs(m - 0.75) < 1e-12, m
hd.barrier(sync_device=False)
hd.finalize()
print("rank", rank, "ok")
"""


def test_shard_range_covers_everything():
    from hehub_amd.dist import shard_range

    for total in (0, 1, 7, 256, 2048, 4097):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]


def _bench(*flags, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_bench_gpus2_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a torch.distributed environment starts two ranks itself;
    here with the launcher self-test (gloo, dummy step): rendezvous, fences, max over ranks, ONE line from rank 0"""
    import json

    p = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-selftest")
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["dist_ranks"] == 2 and r["rccl_ranks"] == 0 and r["steps"] == 3 and r["metric"] == "launcher_selftest"
    assert r["ms_per_step"] >= 2.0      # the dummy step sleeps 2 ms: the timed region really ran `steps` steps


def test_bench_refuses_more_gpus_than_present():
    """fewer devices than --gpus: refuse loudly instead of reporting a one-GPU number as an N-GPU one"""
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    p = _bench("--gpus", str(have + 2), "--steps", "1", "--warmup", "0")
    assert p.returncode == 2 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in p.stderr


def test_bench_rank_refuses_wrong_world_size():
    """a rank whose WORLD_SIZE disagrees with --gpus does not print a line"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
