"""What the host actually gives this process: hardware threads it may run on and its CPU-time quota (cgroup)."""
from __future__ import annotations

import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cpuset_count(text):
    """number of CPUs in a cpuset list such as "0-63,128-191" """
    n = 0
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        n += int(hi or lo) - int(lo) + 1
    return n


def cpu_quota_cores(root="/sys/fs/cgroup"):
    """CPU time this cgroup may use, in cores (float), or None when unlimited / unknown.
    cgroup v2: cpu.max = "<quota> <period>" | "max <period>"; v1: cpu/cpu.cfs_quota_us, cpu/cpu.cfs_period_us"""
    v2 = _read(os.path.join(root, "cpu.max"))
    if v2:
        quota, _, period = v2.partition(" ")
        if quota != "max":
            try:
                return int(quota) / float(period or 100000)
            except ValueError:
                return None
        return None
    q, p = _read(os.path.join(root, "cpu", "cpu.cfs_quota_us")), _read(os.path.join(root, "cpu", "cpu.cfs_period_us"))
    if q and p and int(q) > 0:
        return int(q) / float(p)
    return None


def usable_cores(cap=64, root="/sys/fs/cgroup"):
    """(P, facts): P = how many single-threaded processes can run at full speed at once = min(affinity mask, cpuset,
    CPU quota rounded down, cap); facts says where each number came from"""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cpuset = cpuset_count(_read(os.path.join(root, "cpuset.cpus.effective")) or _read(os.path.join(root, "cpuset", "cpuset.effective_cpus")))
    quota = cpu_quota_cores(root)
    p = affinity
    if cpuset:
        p = min(p, cpuset)
    if quota:
        p = min(p, max(1, int(quota)))
    p = max(1, min(p, cap))
    return p, {"cores_visible": affinity, "cpuset_cores": cpuset or None, "cpu_quota_cores": quota, "cap": cap}
