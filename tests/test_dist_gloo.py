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
