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
    window: Any = None   # (t0, t1) of the most recent timed_launches region

    def sclk(self, window=None):
        """mean shader clock (MHz) of this rank's GPU over a timed window (amdgpu hwmon sampler), None when not sampled"""
        w = window or self.window
        if self.chip is None or w is None:
            return None
        return self.chip.summarize(*w).get("sclk_MHz")


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
    run.window = (t0, t1)
    dt = run.hd.max_over_ranks(t1 - t0, device=run.cdev)
    run.hd.barrier()
    return dt, launches, kern_ms


def family_of_kernel(name):
    """profiling family (hp_prof.cpp) of a kernel name as rocprofv3 prints it (profiles/traffic.json `by_kernel`)"""
    for pat, fam in (("k_ntt_fwd_drop", "ntt_drop"), ("k_ntt_fwd", "ntt"), ("k_ntt_inv", "intt"), ("k_ks_inner", "ks_inner"), ("k_tensor", "tensor")):
        if pat in name:
            return fam
    return None


def step_kernels(run: Run, wl, level, steps=3):
    """Every launch of one step, per profiling family (a separate short pass AFTER the timed regions: hp_prof_begin("*") brackets
    every launch with HIP events): launches per step, average milliseconds, the family's share of A_step and the
    HBM-roofline fraction that makes; plus `step_traffic`: the MEASURED HBM bytes of one step -- the sum over every kernel of the
    step of 2 x FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 counter passes of this very command (tools/prof_step_traffic.sh,
    profiles/traffic.json "step_<workload>_<level>") -- beside the model's A_step / A_min."""
    wl.step()
    run.hd.barrier()
    run.eng.prof_begin("*")
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    run.torch.cuda.synchronize()
    t1 = time.perf_counter()
    fams = run.eng.prof_end_families()
    split = wl.family_limbs()
    S = 8.0 * wl.n
    table, total_ms = {}, 0.0
    for fam, (launches, ms) in fams.items():
        per_step = ms / steps
        total_ms += per_step
        ent = {"launches_per_step": launches / steps, "ms_per_step": per_step, "avg_launch_ms": ms / launches}
        if fam in split:
            b = split[fam] * S * wl.B
            ent.update({"algorithmic_bytes_per_step": b, "achieved_GBps": b / (per_step * 1e-3) / 1e9,
                        "frac_of_hbm_peak": b / (per_step * 1e-3) / 1e9 / HBM_PEAK_GBS})
        table[fam] = ent
    out = {"kernels": table, "kernel_ms_per_step": total_ms, "wall_ms_per_step": 1e3 * (t1 - t0) / steps, "steps": steps,
           "what": "in-library HIP events around EVERY launch of one step (pass after the timed region); algorithmic bytes = the family's share of A_step"}
    try:
        with open(TRAFFIC_FILE) as f:
            tr = json.load(f).get(f"step_{wl.name}_{level}")
    except (OSError, ValueError):
        tr = None
    if tr and tr.get("batch") and wl.n == tr.get("N") and wl.L == tr.get("L"):
        # each family of the table against its own MEASURED bytes as well (the A_step share counts the key once per ciphertext and
        # every digit row at 8 bytes: the inner product's share is 2 x what crosses HBM, its A_step fraction is above 1 for that reason)
        for fam, ent in table.items():
            b = sum(v for k, v in tr.get("by_kernel", {}).items() if family_of_kernel(k) == fam) * wl.B / tr["batch"]
            if b:
                ent.update({"measured_bytes_per_step": b, "measured_GBps": b / (ent["ms_per_step"] * 1e-3) / 1e9,
                            "measured_frac_of_hbm_peak": b / (ent["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
        per_op = tr["bytes_per_step"] / tr["batch"]
        a_step = wl.a_limbs * S
        a_min = (6 * wl.L - 2) * S + 2 * wl.L * (wl.L + 1) * S / wl.B
        ms_step = 1e3 * (t1 - t0) / steps
        out["step_traffic"] = {"measured_bytes_per_op": per_op, "A_step_bytes_per_op": a_step, "A_min_bytes_per_op": a_min,
                               "measured_over_A_step": per_op / a_step, "measured_over_A_min": per_op / a_min,
                               "achieved_GBps": per_op * wl.B / (ms_step * 1e-3) / 1e9,
                               "frac_of_hbm_peak": per_op * wl.B / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "by_kernel_bytes_per_op": {k: v / tr["batch"] for k, v in tr.get("by_kernel", {}).items()},
                               "source": tr.get("source")}
    return out


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


def roofline_entry(family, alg_bytes_per_step, steps, launches, kern_ms, elapsed, logn, spread, level="B", sclk_mhz=None):
    """the `roofline` object of the line: dominant kernel family, HIP events recorded by the library on the launch stream
    around every launch of that family inside the timed region (rank-local); HBM traffic per launch and VALUBusy from the
    committed PMC measurement of this kernel shape (rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE,
    tools/prof_round.sh), null when no measurement exists"""
    bytes_per_launch = alg_bytes_per_step * steps / launches
    avg_s = kern_ms * 1e-3 / launches
    achieved = bytes_per_launch / avg_s / 1e9
    # `bound` names what the committed counters say limits the kernel: "valu" when its vector ALUs are busy most of the time while
    # its HBM traffic is a fraction of the peak (the transforms), "hbm" otherwise.  achieved / peak / frac are the HBM roofline
    # either way (algorithmic bytes over the launch duration); `alu` prices the same launch against the issue peak.
    out = {"bound": "hbm", "priced_against": "hbm", "kernel": KERNEL_NAMES.get(family, family), "achieved": achieved, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": launches,
           "avg_launch_ms": kern_ms / launches, "algorithmic_bytes_per_launch": bytes_per_launch,
           "share_of_step_time": kern_ms * 1e-3 / elapsed}
    try:
        with open(TRAFFIC_FILE) as f:
            trs = json.load(f)
        tr = None
        a = "_a" if level == "A" else ""
        if family == "ntt":   # the digit-spread launch has its own measurement where one exists
            tr = (trs.get(f"k_ntt_fwd{a}_logn{logn}_spread") if spread else None) or trs.get(f"k_ntt_fwd{a}_logn{logn}")
        elif family == "intt":
            tr = trs.get(f"k_ntt_inv{a}_logn{logn}")
        if tr:
            limbs_per_launch = bytes_per_launch / (16.0 * (1 << logn))
            out["traffic"] = tr["bytes_per_limb"] * limbs_per_launch
            out["traffic_source"] = ("rocprofv3 PMC per-limb measurement x limbs per launch "
                                     f"(profiles/traffic.json: {tr.get('source', 'see _comment')})")
            if "valu_busy" in tr:
                out["valu_busy"] = tr["valu_busy"]
            traffic_frac = out["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS
            out["traffic_frac_of_hbm_peak"] = traffic_frac
            if tr.get("valu_busy", 0) >= 0.6 and traffic_frac < 0.5:
                out["bound"] = "valu"
            if "valu_insts_per_wave" in tr:
                # the committed instruction count of this kernel shape against THIS run's launch duration and shader clock:
                # N / 2048 waves per limb transform (32 coefficients per thread, hp_ntt_tile.h Geo: 16 waves at N = 32768), 1024 SIMDs
                # (256 CUs x 4), one VALU instruction of the kernel's mix occupies its SIMD for cycles_per_inst_mix cycles
                # (4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU; 4 = a full-rate instruction)
                waves = limbs_per_launch * max(1, (1 << logn) // 2048)
                cpi = tr.get("valu_cycles_per_inst", 4.0)
                alu = {"valu_insts_per_wave": tr["valu_insts_per_wave"], "waves": waves, "cycles_per_inst_mix": cpi, "sclk_MHz": sclk_mhz,
                       "simds": 1024}
                if sclk_mhz:
                    alu["frac_of_issue_peak"] = tr["valu_insts_per_wave"] * waves * cpi / (1024 * sclk_mhz * 1e6 * avg_s)
                out["alu"] = alu
    except (OSError, ValueError, KeyError):
        pass
    return out
