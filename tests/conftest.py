"""pytest configuration: markers and shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, oracle vs the
compiled reference (when oracle/_ref exists), host logic, C-ABI symbol checks.
`-m gpu` runs on an MI355X: parity of the HIP engine against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Host <-> device copies of numpy arrays through torch (the tests' to_device / to_host): by default the HIP runtime pins the
# pageable numpy pages in place for every copy of 1 MiB or more (hsa_amd_memory_lock on memory numpy frees and re-maps all the
# time) -- about one run in eight of this suite died of a GPU page fault inside such a copy (rocgdb: VMFaultHandler while the main
# thread sat in DmaBlitManager::hsaCopyStagedOrPinned -> addPinnedMem; NOTES.md).  The staging path has no such window.  Read when
# the runtime starts, inherited by the subprocesses the tests spawn; the engine's own transfers use memory it registers itself.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")   # MiB


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    config.addinivalue_line("markers", "perf: compares measured rates / times with thresholds or with each other (needs an MI355X); NOT part of "
                                       "the correctness tiers: deselected unless the -m expression names perf (`-m perf`)")


@pytest.fixture(scope="session")
def orc():
    from oracle.pyoracle import Oracle, build

    build(ref=True)
    return Oracle("orc")


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Oracle, build, have_ref

    build(ref=True)
    if not have_ref():
        pytest.skip("oracle/_ref/libhehub_ref.so not built (no /root/reference here)")
    return Oracle("ref")


def pytest_collection_modifyitems(config, items):
    # "slow" is informational only; everything not marked gpu runs on CPU.
    # "perf" tests hold wall-clock thresholds and ratios: a noisy box must not turn a parity-green tier red (and, with -x, hide every later
    # test), so they run only when asked for by name: `pytest -m perf`.
    if "perf" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("perf") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
