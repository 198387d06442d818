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


def parse_cpulist(text):
    """CPUs of a list such as "0-3,8,10-11" as a set of ints"""
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(device, lookup=None):
    """Bind this process (one rank of bench.py) to the CPUs of the NUMA node its GPU hangs off -- launches and the staging copies of the
    verify pass then run on that socket (an 8-GPU node has two).  `lookup(device) -> (numa_node, cpulist)`; default: the engine's
    hp_device_numa (PCI address of the HIP device -> sysfs).  HP_BENCH_NO_AFFINITY leaves the process alone.  Returns what was done."""
    out = {"numa_node": None, "cpus_bound": 0, "cpus_before": len(os.sched_getaffinity(0))}
    if os.environ.get("HP_BENCH_NO_AFFINITY"):
        out["skipped"] = "HP_BENCH_NO_AFFINITY"
        return out
    try:
        if lookup is None:
            import ctypes as C

            from hehub_amd import capi

            lib = capi.load()
            node, buf = C.c_int(-1), C.create_string_buffer(1024)
            lib.hp_device_numa(int(device), C.byref(node), buf, 1024)
            numa, cpulist = node.value, buf.value.decode()
        else:
            numa, cpulist = lookup(device)
        out["numa_node"] = numa
        want = parse_cpulist(cpulist) & os.sched_getaffinity(0)
        if numa is not None and numa >= 0 and want:
            os.sched_setaffinity(0, want)
            out["cpus_bound"] = len(want)
            out["cpulist"] = cpulist
    except Exception as e:   # placement is an optimisation, never a reason to fail a run
        out["error"] = repr(e)[:200]
    return out
