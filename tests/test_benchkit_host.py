"""benchkit/host.py: how many single-threaded CPU-baseline processes this host can really run at once (VERDICT r02 item 8:
the P of cpu_baseline_node must respect the cgroup CPU quota, not only the affinity mask)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchkit import host  # noqa: E402


def _cg(tmp_path, **files):
    for name, text in files.items():
        p = tmp_path / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    return str(tmp_path)


def test_cpuset_lists():
    assert host.cpuset_count("0-63,128-191") == 128 and host.cpuset_count("5") == 1 and host.cpuset_count("0,2,4-5\n") == 4
    assert host.cpuset_count("") == 0 and host.cpuset_count(None) == 0


def test_quota_v2_and_v1(tmp_path):
    assert host.cpu_quota_cores(_cg(tmp_path / "a", **{"cpu.max": "1500000 100000\n"})) == 15.0
    assert host.cpu_quota_cores(_cg(tmp_path / "b", **{"cpu.max": "max 100000\n"})) is None
    assert host.cpu_quota_cores(_cg(tmp_path / "c", **{"cpu/cpu.cfs_quota_us": "250000", "cpu/cpu.cfs_period_us": "100000"})) == 2.5
    assert host.cpu_quota_cores(_cg(tmp_path / "d", **{"cpu/cpu.cfs_quota_us": "-1", "cpu/cpu.cfs_period_us": "100000"})) is None
    assert host.cpu_quota_cores(str(tmp_path / "missing")) is None


def test_usable_cores_takes_the_minimum(tmp_path):
    aff = len(os.sched_getaffinity(0))
    p, facts = host.usable_cores(64, _cg(tmp_path / "a", **{"cpu.max": "300000 100000", "cpuset.cpus.effective": "0-255"}))
    assert p == min(3, aff) and facts["cpu_quota_cores"] == 3.0 and facts["cpuset_cores"] == 256 and facts["cores_visible"] == aff
    p, _ = host.usable_cores(64, _cg(tmp_path / "b", **{"cpu.max": "50000 100000"}))     # half a core still runs one process
    assert p == 1
    p, facts = host.usable_cores(2, str(tmp_path / "none"))
    assert p == min(2, aff) and facts["cpu_quota_cores"] is None
