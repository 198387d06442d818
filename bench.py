#!/usr/bin/env python3
"""Headline benchmark of the MI355X ring-arithmetic engine.

    python bench.py --gpus 1 --steps K --warmup W [--workload ckks|ntt|ntt15|intt|intt15|bgv|rotate] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic ciphertexts that is already
resident in HBM:

  ckks (default, BASELINE config 3/4): ckks::mult + relinearize + rescale_inplace on `batch`
        ciphertext pairs, N = 32768, L = 10 moduli {50,40x9} bits + 50-bit special prime
  ntt  (BASELINE config 2): forward negacyclic NTT of 1024 polynomials x 4 limbs, N = 16384
  bgv  (BASELINE config 5 shape): bgv mult + relinearize + mod_switch, N = 8192, L = 6

One process per GPU; ciphertext batches are sharded across ranks with no data-path collective
(SURVEY.md section 8e), so per-GPU work is fixed: weak scaling.  Rank 0 prints ONE JSON line.

The oracle / compiled reference is used only for the `cpu_baseline` leg (rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the host driver of these boxes only supports dmabuf IPC: RCCL needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ckks", choices=["ckks", "ntt", "ntt15", "intt", "intt15", "bgv", "rotate", "ckks-limb", "encdec", "mul", "add", "ckks-hks"])
    ap.add_argument("--batch", type=int, default=0, help="units per GPU per step (0 = BASELINE config value)")
    ap.add_argument("--logn", type=int, default=0,
                    help="override log2 of the ring degree for the ntt/intt/ckks/rotate workloads (same moduli); "
                         "0 = the BASELINE config's N")
    ap.add_argument("--ntt-rates", action="store_true",
                    help="ckks workload: also time forward / inverse limb transforms of the same batch (outside the timed "
                         "region) and report them under \"ntt\".  Off by default so that a rocprofv3 summary of the default "
                         "command shows k_ntt_fwd in its digit-spread launches only, as the roofline object does")
    ap.add_argument("--hks-alpha", type=int, default=2, help="ckks-hks: ciphertext moduli per key-switch digit")
    ap.add_argument("--hks-k", type=int, default=2, help="ckks-hks: number of (50-bit) special primes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline sample")
    ap.add_argument("--cpu-procs", type=int, default=1,
                    help="cpu_baseline as this many independent processes at once (the reference has process-global "
                         "unsynchronised caches, so processes, not threads); the rates are summed, cores = this number")
    return ap.parse_args()


def rand_words(torch, shape, moduli, device, seed):
    """uniform words in [0, q_k) per limb (limb axis = -2), generated on the device"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(shape, dtype=torch.int64, device=device)
    for k, q in enumerate(moduli):
        out.select(-2, k).copy_(torch.randint(0, int(q), out.select(-2, k).shape, generator=g, device=device,
                                              dtype=torch.int64))
    return out


LOGN_OVERRIDE = 0   # set from --logn so that the CPU leg times the same ring degree


def cpu_baseline(workload, P, budget_s):
    """Time the CPU path on this host: the compiled reference when oracle/_ref travelled with the
    snapshot ("reference"), else the C restatement ("port").  Single thread, bounded sample."""
    import numpy as np
    from oracle.pyoracle import Oracle, SplitMix, have_ref, build

    build(ref=False)
    kind = "reference" if have_ref() else "port"
    lib = Oracle("ref" if kind == "reference" else "orc")
    rng = SplitMix(3)
    if workload in ("mul", "add"):
        q, n = P.C2_MODULI[0], 1 << P.C2_LOGN
        a, b = rng.words(n, q), rng.words(n, q)
        f = (lambda: lib.mul_hybrid_lazy(q, a, b)) if workload == "mul" else (lambda: lib.poly_add([q], a[None], b[None]))
        f()
        iters, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            f(); iters += 1
        per = (time.perf_counter() - t0) / iters
        return {"value": 1.0 / per, "unit": "limb-op/s", "cores": 1, "kind": kind,
                "sample": f"{iters} x one-limb {workload}, N={n}, q={q}, single thread (includes the ctypes call and one result allocation)"}
    if "ntt" in workload:
        logn, q = (P.C3_LOGN, P.C3_P) if workload.endswith("15") else (P.C2_LOGN, P.C2_MODULI[0])
        logn = LOGN_OVERRIDE or logn
        inv = int(workload.startswith("intt"))
        x = rng.words(1 << logn, q)
        if kind == "reference":
            per = lib.time_ntt(logn, q, inv, 200, x)
            iters = max(200, int(budget_s / max(per, 1e-6)))
            per = lib.time_ntt(logn, q, inv, iters, x)
        else:
            f = lib.intt if inv else lib.ntt
            f(logn, q, x)
            iters, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s:
                f(logn, q, x); iters += 1
            per = (time.perf_counter() - t0) / iters
        return {"value": 1.0 / per, "unit": "limb-NTT/s", "cores": 1, "kind": kind,
                "sample": f"{iters} {'inverse' if inv else 'forward'} NTTs of one limb, N={1 << logn}, q={q}, single thread, tables warm"}
    if workload == "encdec":
        logn, moduli = P.C3_LOGN, P.C3_Q
        noise, c1, pt, sk = P.edge_case(rng, logn, moduli)
        f = lambda: lib.rlwe_decrypt_core(moduli, lib.rlwe_encrypt_core(moduli, noise, c1, pt, sk), sk)
        f()
        iters, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            f(); iters += 1
        per = (time.perf_counter() - t0) / iters
        return {"value": 1.0 / per, "unit": "ciphertext/s", "cores": 1, "kind": kind,
                "sample": f"{iters} x (encrypt_core on given samples + decrypt_core) of one ciphertext, N={1 << logn}, L={len(moduli)}, single thread, tables warm"}
    if workload == "rotate":   # the reference's own benchmark workload (bench/benchmarks.cpp:21-37) at the C3 shape
        logn, mext = LOGN_OVERRIDE or P.C3_LOGN, P.C3_MODULI_EXT
        n, L = 1 << logn, len(mext) - 1
        ct = rng.poly((2, L, n), mext[:L])
        key = rng.poly((L, 2, L + 1, n), mext)
        lib.ckks_rotate(mext, ct, key, 1)
        iters, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            lib.ckks_rotate(mext, ct, key, 1); iters += 1
        per = (time.perf_counter() - t0) / iters
        return {"value": 1.0 / per, "unit": "rotation/s", "cores": 1, "kind": kind,
                "sample": f"{iters} x ckks::rotate(ct, key, 1) on one ciphertext, N={n}, L={L}, single thread, tables warm"}
    if workload == "ckks":
        logn, mext, t = LOGN_OVERRIDE or P.C3_LOGN, P.C3_MODULI_EXT, 0
    else:
        logn, mext, t = P.C5_LOGN, P.C5_MODULI_EXT, P.C5_T
    n, L = 1 << logn, len(mext) - 1
    ct1 = rng.poly((2, L, n), mext[:L]); ct2 = rng.poly((2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    if kind == "reference":
        f = (lambda it: lib.time_ckks_mult(mext, ct1, ct2, key, it)) if workload == "ckks" else \
            (lambda it: lib.time_bgv_mult(mext, t, ct1, ct2, key, it))
        per = f(2)
        iters = max(2, int(budget_s / max(per, 1e-6)))
        per = f(iters)
    else:
        f = (lambda: lib.ckks_mult(mext, ct1, ct2, key)) if workload == "ckks" else \
            (lambda: lib.bgv_mult(mext, t, ct1, ct2, key))
        f()
        iters, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            f(); iters += 1
        per = (time.perf_counter() - t0) / iters
    name = "ckks::mult+relinearize+rescale_inplace" if workload == "ckks" else "bgv mult+relinearize+mod_switch"
    return {"value": 1.0 / per, "unit": "hom-mult/s", "cores": 1, "kind": kind,
            "sample": f"{iters} x {name} on one ciphertext pair, N={n}, L={L}, single thread, tables warm"}


def _cpu_baseline_worker(workload, budget_s, logn_override=0):
    import params as P

    global LOGN_OVERRIDE
    LOGN_OVERRIDE = logn_override
    return cpu_baseline(workload, P, budget_s)


def main():
    global LOGN_OVERRIDE
    args = parse()
    LOGN_OVERRIDE = args.logn
    import torch

    import params as P
    from hehub_amd.engine import Engine

    from hehub_amd import dist as hd

    world, rank, local = hd.env_world()
    torch.cuda.set_device(local)
    hd.init("nccl", device=torch.device(f"cuda:{local}"))   # "nccl" is RCCL on ROCm; rendezvous + timing fences only
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = f"cuda:{local}"
    eng = Engine(local)

    wl = args.workload
    if wl in ("ntt", "ntt15", "intt", "intt15"):
        inverse = wl.startswith("intt")
        if wl.endswith("15"):   # the transform shape inside the C3 pipeline: N=32768, 10 moduli + special prime
            logn, moduli = P.C3_LOGN, P.C3_MODULI_EXT
            B = args.batch or 256
        else:
            logn, moduli = P.C2_LOGN, P.C2_MODULI
            B = args.batch or P.C2_BATCH
        logn = args.logn or logn
        n, L = 1 << logn, len(moduli)
        x = rand_words(torch, (B, L, n), moduli, dev, 2 + rank)
        units_per_step = B * L
        step = (lambda: eng.intt_(moduli, x)) if inverse else (lambda: eng.ntt_(moduli, x))
        family = "intt" if inverse else "ntt"
        alg_bytes_per_step = 16.0 * n * B * L
        launches_per_step = 1
        metric, unit = "limb_ntt_per_s", "limb-NTT/s"
        cfg = {"workload": f"{'C2' if (logn == 14 and L == 4) else 'C3-shape' if logn == 15 else 'custom'}: batched {'inverse' if inverse else 'forward'} negacyclic NTT, N={n}, {L} RNS limbs, batch={B} polynomials per GPU",
               "N": n, "limbs": L, "batch_per_gpu": B}
    elif wl in ("mul", "add"):
        # coefficient-wise kernels at the C2 shape: RnsPolynomial operator* (hybrid Montgomery+Harvey product,
        # rns.cpp:120-140) / operator+= (rns.cpp:58-87); 24*N algorithmic bytes per limb
        logn, moduli = P.C2_LOGN, P.C2_MODULI
        B = args.batch or P.C2_BATCH
        n, L = 1 << logn, len(moduli)
        a = rand_words(torch, (B, L, n), moduli, dev, 21 + rank)
        b = rand_words(torch, (B, L, n), moduli, dev, 22 + rank)
        out = eng.empty((B, L, n))
        units_per_step = B * L
        step = (lambda: eng.poly_mul(moduli, a, b, out=out)) if wl == "mul" else (lambda: eng.poly_add(moduli, a, b, out=out))
        family = "elem"
        alg_bytes_per_step = 24.0 * n * B * L
        launches_per_step = 1
        metric, unit = f"limb_{wl}_per_s", "limb-op/s"
        cfg = {"workload": f"C2 shape: coefficient-wise modular {'multiply' if wl == 'mul' else 'add'}, N={n}, {L} RNS limbs, batch={B} polynomials per GPU",
               "N": n, "limbs": L, "batch_per_gpu": B}
    elif wl == "ckks-hks":
        # EXTENSION, not comparable with the reference: the C3 ciphertext chain with a hybrid key switch (digits of
        # --hks-alpha moduli, --hks-k special primes); keys in the hybrid format, results differ from hehub's by design
        logn = args.logn or P.C3_LOGN
        L, k, alpha = len(P.C3_Q), args.hks_k, args.hks_alpha
        mext = P.C3_Q + P.ntt_primes(k, P.C3_LOGN, 50, exclude=P.C3_Q)
        B = args.batch or P.C3_BATCH
        n, nd = 1 << logn, (L + alpha - 1) // alpha
        ct1 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 3 + rank)
        ct2 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 1003 + rank)
        key = rand_words(torch, (nd, 2, L + k, n), mext, dev, 7)
        out = eng.empty((B, 2, L - 1, n))
        units_per_step = B
        step = lambda: eng.ckks_mult_hks(mext, k, alpha, ct1, ct2, key, out=out)
        family = "ntt"
        fwd = nd * (L + k) - L                       # lifted-digit transforms per ciphertext: the one k_ntt_fwd launch per step
        alg_bytes_per_step = 16.0 * n * fwd * B      # (ModDown and rescale transforms are k_ntt_fwd_drop launches, family "ntt_drop")
        launches_per_step = 1
        metric, unit = "ckks_hks_hom_mult_per_s", "hom-mult/s"
        cfg = {"workload": f"EXTENSION (not hehub-compatible keys): ckks mult + hybrid-key relinearisation (digits of {alpha} moduli, "
                           f"{k} special primes) + rescale, N={n}, L={L}, batch={B} ciphertext pairs per GPU",
               "N": n, "L": L, "batch_per_gpu": B, "hks_alpha": alpha, "hks_k": k, "digits": nd,
               "forward_transforms_per_op": fwd + 2 * L + 2 * (L - 1), "reference_algorithm_forward_transforms_per_op": L * L + 4 * L - 2}
    elif wl == "encdec":
        # either side of the path: encrypt_core on caller-supplied samples, then decrypt_core
        logn, moduli = P.C3_LOGN, P.C3_Q
        B = args.batch or P.C3_BATCH
        n, L = 1 << logn, len(moduli)
        c1 = rand_words(torch, (B, L, n), moduli, dev, 11 + rank)
        pt = rand_words(torch, (B, L, n), moduli, dev, 12 + rank)
        sk = rand_words(torch, (L, n), moduli, dev, 13)
        noise = torch.randint(-19, 20, (B, n), dtype=torch.int64, device=dev)
        units_per_step = B
        step = lambda: eng.rlwe_decrypt_core(moduli, eng.rlwe_encrypt_core(moduli, noise, c1, pt, sk), sk)
        family = "ntt"
        alg_bytes_per_step = 16.0 * n * 2 * L * B      # the two forward transforms per ciphertext (noise, plaintext)
        launches_per_step = 2
        metric, unit = "rlwe_encrypt_decrypt_per_s", "ciphertext/s"
        cfg = {"workload": f"C3 shape: rlwe encrypt_core (given samples) + decrypt_core, N={n}, L={L}, batch={B} ciphertexts per GPU",
               "N": n, "L": L, "batch_per_gpu": B, "A_step_bytes_per_op": (15 * L + 1) * 8 * n}
    else:
        if wl in ("ckks", "rotate", "ckks-limb"):
            logn, mext, t, B0 = P.C3_LOGN, P.C3_MODULI_EXT, 0, (P.C3_BATCH if wl != "ckks-limb" else 8)
        else:
            logn, mext, t, B0 = P.C5_LOGN, P.C5_MODULI_EXT, P.C5_T, 512
        B = args.batch or B0
        if wl in ("ckks", "rotate"):
            logn = args.logn or logn
        n, L = 1 << logn, len(mext) - 1
        ct1 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 3 + rank)
        ct2 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 1003 + rank)
        key = rand_words(torch, (L, 2, L + 1, n), mext, dev, 7)
        out = eng.empty((B, 2, L - 1, n))
        units_per_step = B
        # compulsory bytes per op in limbs (S = 8N): SURVEY.md 8d for hom-mult; for a rotation the tensor product
        # (7L) becomes the gather (4L), there is no second drop and only c0 gets the moved addend
        a_limbs = 5 * L * L + 36 * L
        # the dominant kernel is k_ntt_fwd in its digit-spread launch (rgsw.cpp:108-119): L*L limb transforms per
        # ciphertext, one launch per step; the fused drop-last-prime launches are a different kernel (k_ntt_fwd_drop,
        # profiling family "ntt_drop") and are not mixed into this roofline
        fwd_per_ct = L * L
        scaling = "weak"
        if wl == "ckks-limb":
            # latency mode (hehub_amd/sharded.py): the SAME small batch on every rank, cut by output modulus, with
            # the all-gather of the key-switch digits over RCCL; total work is fixed as N grows -> strong scaling
            from hehub_amd.sharded import Comm, ShardedMult

            comm, sm = Comm(), ShardedMult(eng, mext, world)
            ct1 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 3)
            ct2 = rand_words(torch, (B, 2, L, n), mext[:L], dev, 1003)
            bufs = sm.buffers(B, n)
            step = lambda: sm.run(comm, ct1, ct2, key, bufs)
            metric, unit = "ckks_hom_mult_per_s", "hom-mult/s"
            name = "C3 shape, limb-sharded latency mode: ckks::mult + relinearize + rescale_inplace"
            scaling = "strong"
            units_per_step = B / world   # `value` multiplies by world below: the batch is shared, not replicated work
        elif wl == "rotate":
            step = lambda: eng.ckks_rotate(mext, ct1, key, 1)
            metric, unit = "ckks_rotation_per_s", "rotation/s"
            name = "C3 shape: ckks::rotate (gather + key switch + drop of the special prime)"
            a_limbs = 5 * L * L + 22 * L + 6
        elif wl == "ckks":
            step = lambda: eng.ckks_mult(mext, ct1, ct2, key, out=out)
            metric, unit = "ckks_hom_mult_per_s", "hom-mult/s"
            name = "C3: ckks::mult + relinearize + rescale_inplace"
        else:
            step = lambda: eng.bgv_mult(mext, t, ct1, ct2, key, out=out)
            metric, unit = "bgv_hom_mult_per_s", "hom-mult/s"
            name = "C5 shape: bgv mult_low_level + relinearize + mod_switch_inplace"
        family = "ntt"
        alg_bytes_per_step = 16.0 * n * fwd_per_ct * B
        launches_per_step = None
        cfg = {"workload": f"{name}, N={n}, L={L} moduli + special prime, batch={B} ciphertext pairs per GPU",
               "N": n, "L": L, "batch_per_gpu": B, "sub_batch": int(os.environ.get("HP_MULT_CHUNK", "0")) or B,
               "A_step_bytes_per_op": a_limbs * 8 * n}

    for _ in range(args.warmup):
        step()
    hd.barrier()                       # dist.barrier() + torch.cuda.synchronize()
    eng.prof_begin(family)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    launches, kern_ms = eng.prof_end()
    elapsed = hd.max_over_ranks(t1 - t0, device=dev)
    hd.barrier()

    value = units_per_step * world * args.steps / elapsed
    res = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling if wl == "ckks-limb" else "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "config": cfg,
    }
    # roofline of the dominant kernel family (forward NTT), from HIP events recorded by the library on the
    # launch stream around every launch of that family inside the timed region (rank-local)
    if launches:
        bytes_per_launch = alg_bytes_per_step * args.steps / launches
        avg_s = kern_ms * 1e-3 / launches
        achieved = bytes_per_launch / avg_s / 1e9
        kname = {"intt": "k_ntt_inv (register/LDS-tiled inverse NTT)", "elem": "k_poly_binary (coefficient-wise)"}.get(
            family, "k_ntt_fwd (register/LDS-tiled forward NTT)")
        res["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None, "launches": launches, "avg_launch_ms": kern_ms / launches,
                           "algorithmic_bytes_per_launch": bytes_per_launch,
                           "share_of_step_time": kern_ms * 1e-3 / elapsed}
        # HBM traffic per launch from the committed PMC measurement of this kernel shape (rocprofv3 --pmc
        # FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, tools/prof_pmc.sh); null when no measurement exists
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                trs = json.load(f)
                tr = None
                if family == "ntt":   # the digit-spread launch has its own measurement where one exists
                    tr = (trs.get(f"k_ntt_fwd_logn{logn}_spread") if wl in ("ckks", "bgv", "rotate") else None) or trs.get(f"k_ntt_fwd_logn{logn}")
            if tr:
                limbs_per_launch = bytes_per_launch / (16.0 * n)
                res["roofline"]["traffic"] = tr["bytes_per_limb"] * limbs_per_launch
                res["roofline"]["traffic_source"] = "rocprofv3 PMC per-limb measurement x limbs per launch (profiles/r01_traffic.json)"
        except (OSError, ValueError):
            pass
    if wl in ("ckks", "bgv", "rotate", "ckks-limb"):
        a_step = a_limbs * 8 * n
        res["pipeline_roofline"] = {"A_step_GBps": value / world * a_step / 1e9,
                                    "frac_of_hbm_peak": value / world * a_step / 1e9 / HBM_PEAK_GBS}
        if wl in ("ckks", "bgv"):   # the two other yardsticks of SURVEY.md 8d: every primitive its own pass / I-O lower bound
            a_prim = (6 * L * L + 67 * L) * 8 * n
            a_min = (6 * L - 2) * 8 * n + 2 * L * (L + 1) * 8 * n / B
            res["pipeline_roofline"].update({"A_prim_frac_of_hbm_peak": value / world * a_prim / 1e9 / HBM_PEAK_GBS,
                                             "A_min_frac_of_hbm_peak": value / world * a_min / 1e9 / HBM_PEAK_GBS})
    if wl == "ckks" and args.ntt_rates:
        # BASELINE.json's metric names both rates ("NTT/s and CKKS hom-mult/s ... N=32768"): the line also carries the
        # limb-transform rates at the same ring degree (all limbs of the same batch of ciphertexts, forward and inverse
        # timed separately, same fences; outside the timed region of `value`)
        xq = ct1.view(B * 2, L, n)
        rates = {}
        for name, fn in (("forward", lambda: eng.ntt_(mext[:L], xq)), ("inverse", lambda: eng.intt_(mext[:L], xq))):
            fn()
            hd.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            dt = hd.max_over_ranks(time.perf_counter() - t0, device=dev)
            hd.barrier()
            rates[name] = B * 2 * L * world * args.steps / dt
        res["ntt"] = {"N": n, "limbs_per_launch": B * 2 * L, "forward_limb_ntt_per_s": rates["forward"],
                      "inverse_limb_ntt_per_s": rates["inverse"],
                      "forward_frac_of_hbm_peak": rates["forward"] / world * 16.0 * n / 1e9 / HBM_PEAK_GBS,
                      "inverse_frac_of_hbm_peak": rates["inverse"] / world * 16.0 * n / 1e9 / HBM_PEAK_GBS}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                cwl = "ckks" if wl in ("ckks-limb", "ckks-hks") else wl
                if args.cpu_procs > 1:
                    import multiprocessing as mp

                    with mp.get_context("spawn").Pool(args.cpu_procs) as pool:
                        parts = pool.starmap(_cpu_baseline_worker, [(cwl, args.cpu_seconds, args.logn)] * args.cpu_procs)
                    res["cpu_baseline"] = dict(parts[0], value=sum(p["value"] for p in parts), cores=args.cpu_procs,
                                               sample=f"{args.cpu_procs} concurrent processes, each: " + parts[0]["sample"])
                else:
                    res["cpu_baseline"] = cpu_baseline(cwl, P, args.cpu_seconds)
            except Exception as e:  # the checker is optional infrastructure; the GPU number stands on its own
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    hd.finalize()
    eng.close()


if __name__ == "__main__":
    main()
