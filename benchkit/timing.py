"""Fences, in-library kernel events and the roofline arithmetic shared by the timed region and the extra sections."""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from typing import Any

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); the measured copy ceiling is reported next to it
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # PMC measurements per kernel shape (tools/traffic_from_pmc.py)


@dataclass
class Run:
    """what every part of a bench run needs: torch, the rank's engine, the rendezvous helpers, the parameter sets"""
    torch: Any
    hd: Any        # hehub_amd.dist
    eng: Any       # hehub_amd.engine.Engine
    P: Any         # tests/params.py
    args: Any
    world: int
    rank: int
    dev: str       # "cuda:<local>"
    cdev: str      # where the tensors of the few collectives live ("cpu" in the shared-GPU test mode)
    chip: Any = None   # benchkit.chip.ChipSampler of this rank's GPU


def timed_launches(run: Run, fn, family, steps, warm=1):
    """`steps` calls of fn between fences; returns (wall seconds max over ranks, launches of `family`, their kernel ms from
    the library's HIP events on its stream)"""
    for _ in range(warm):
        fn()
    run.hd.barrier()
    run.eng.prof_begin(family)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    run.torch.cuda.synchronize()
    t1 = time.perf_counter()
    launches, kern_ms = run.eng.prof_end()
    dt = run.hd.max_over_ranks(t1 - t0, device=run.cdev)
    run.hd.barrier()
    return dt, launches, kern_ms


def rate_entry(units_per_launch, bytes_per_unit, steps, world, dt, launches, kern_ms):
    """units/s over the wall clock of the region (whole job) and the achieved GB/s of the kernel itself (events)"""
    e = {"per_s": units_per_launch * world * steps / dt,
         "wall_GBps_per_gpu": units_per_launch * bytes_per_unit * steps / dt / 1e9}
    if launches:
        gbps = units_per_launch * bytes_per_unit * steps / (kern_ms * 1e-3) / 1e9
        e.update({"avg_launch_ms": kern_ms / launches, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS})
    return e


KERNEL_NAMES = {"intt": "k_ntt_inv (register/LDS-tiled inverse NTT)", "elem": "k_poly_binary (coefficient-wise)",
                "ntt": "k_ntt_fwd (register/LDS-tiled forward NTT)", "copy": "k_copy (stream copy)"}


def roofline_entry(family, alg_bytes_per_step, steps, launches, kern_ms, elapsed, logn, spread):
    """the `roofline` object of the line: dominant kernel family, HIP events recorded by the library on the launch stream
    around every launch of that family inside the timed region (rank-local); HBM traffic per launch and VALUBusy from the
    committed PMC measurement of this kernel shape (rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE,
    tools/prof_round.sh), null when no measurement exists"""
    bytes_per_launch = alg_bytes_per_step * steps / launches
    avg_s = kern_ms * 1e-3 / launches
    achieved = bytes_per_launch / avg_s / 1e9
    out = {"bound": "hbm", "kernel": KERNEL_NAMES.get(family, family), "achieved": achieved, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": launches,
           "avg_launch_ms": kern_ms / launches, "algorithmic_bytes_per_launch": bytes_per_launch,
           "share_of_step_time": kern_ms * 1e-3 / elapsed}
    try:
        with open(TRAFFIC_FILE) as f:
            trs = json.load(f)
        tr = None
        if family == "ntt":   # the digit-spread launch has its own measurement where one exists
            tr = (trs.get(f"k_ntt_fwd_logn{logn}_spread") if spread else None) or trs.get(f"k_ntt_fwd_logn{logn}")
        elif family == "intt":
            tr = trs.get(f"k_ntt_inv_logn{logn}")
        if tr:
            limbs_per_launch = bytes_per_launch / (16.0 * (1 << logn))
            out["traffic"] = tr["bytes_per_limb"] * limbs_per_launch
            out["traffic_source"] = ("rocprofv3 PMC per-limb measurement x limbs per launch "
                                     f"(profiles/traffic.json: {tr.get('source', 'see _comment')})")
            if "valu_busy" in tr:
                out["valu_busy"] = tr["valu_busy"]
    except (OSError, ValueError, KeyError):
        pass
    return out
