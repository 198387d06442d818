#!/usr/bin/env python3
"""Headline benchmark of the MI355X ring-arithmetic engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ckks|ntt|ntt15|intt|intt15|bgv|rotate|...] [--batch B]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`),
after checking that N devices exist; under an external `torch.distributed.run` it is one of the ranks.

A "step" is one pass of the hot path over one batch of synthetic ciphertexts that is already resident in HBM:

  ckks (default, BASELINE config 3/4): ckks::mult + relinearize + rescale_inplace on `batch`
        ciphertext pairs, N = 32768, L = 10 moduli {50,40x9} bits + 50-bit special prime
  ntt  (BASELINE config 2): forward negacyclic NTT of 1024 polynomials x 4 limbs, N = 16384
  bgv  (BASELINE config 5 shape): bgv mult + relinearize + mod_switch, N = 8192, L = 6

One process per GPU; ciphertext batches are sharded across ranks with no data-path collective
(SURVEY.md section 8e), so per-GPU work is fixed: weak scaling.  Rank 0 prints ONE JSON line:

  value / ms_per_step   whole-job throughput of the timed region (barrier + synchronize on both sides, max over ranks)
  roofline              dominant kernel (digit-spread k_ntt_fwd launch), HIP events recorded by the library on its stream
  verified              every output of the timed buffers compared with the CPU checker AFTER the timed region: the batch
                        is periodic (ciphertext i = class i mod 3 of three distinct random inputs, --input-period), so
                        three checker evaluations cover all of them; a mismatch makes the run fail (exit code 1)
  ntt / coeffwise       (default ckks line) the other half of BASELINE's metric: forward / inverse limb-NTT/s at N = 32768
                        and across N = 4096..32768, coefficient-wise multiply / add, each with achieved GB/s and the
                        fraction of HBM peak; timed outside the hom-mult region
  ckks_by_N             (default ckks line) hom-mult/s at N = 4096 .. 32768 (same chain, batch 256), each verified
  cpu_baseline          the compiled reference (or the C restatement) on ONE core of this host; cpu_baseline_node: the
                        same as P independent processes (P stated)
  --roofline-only       drops everything after the timed region (for clean rocprofv3 summaries of the default command)

The oracle / compiled reference is used only after the timed region, as the checker and as the CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# the host driver of these boxes only supports dmabuf IPC: RCCL needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # PMC measurements per kernel shape (tools/prof_pmc.sh)
WORKLOADS = ["ckks", "ntt", "ntt15", "intt", "intt15", "bgv", "rotate", "ckks-limb", "encdec", "mul", "add", "ckks-hks"]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ckks", choices=WORKLOADS)
    ap.add_argument("--batch", type=int, default=0, help="units per GPU per step (0 = BASELINE config value)")
    ap.add_argument("--logn", type=int, default=0,
                    help="override log2 of the ring degree for the ntt/intt/ckks/rotate workloads (same moduli); "
                         "0 = the BASELINE config's N")
    ap.add_argument("--input-period", type=int, default=3,
                    help="the batch repeats this many distinct random inputs (item i = class i mod period) so that EVERY "
                         "output can be checked against the CPU checker; 0 = all items distinct (then --verify samples)")
    ap.add_argument("--no-verify", action="store_true", help="skip the output check after the timed region")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the timed region and its roofline: no transform / coefficient-wise rates, no verification, no "
                         "CPU legs -- a rocprofv3 summary of this command shows k_ntt_fwd in its digit-spread launches only")
    ap.add_argument("--no-rates", action="store_true", help="ckks workload: skip the \"ntt\" / \"coeffwise\" sections")
    ap.add_argument("--hks-alpha", type=int, default=2, help="ckks-hks: ciphertext moduli per key-switch digit")
    ap.add_argument("--hks-k", type=int, default=2, help="ckks-hks: number of (50-bit) special primes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU budget of the cpu_baseline sample")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="cpu_baseline_node: this many independent processes at once (the reference has process-global "
                         "unsynchronised caches, so processes, not threads); -1 = the cores this process may run on "
                         "(at most 64), 0 = skip")
    ap.add_argument("--cpu-node-seconds", type=float, default=6.0, help="CPU budget per process of cpu_baseline_node")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="exercise only the launcher / rendezvous / timing fences with the gloo backend and a dummy step "
                         "(no engine, no GPU); the line says so and carries no throughput claim")
    return ap.parse_args(argv)


# =====================================================================================================================
# launcher
# =====================================================================================================================
def self_launch(args) -> int:
    """--gpus N > 1 without a torch.distributed environment: become the launcher of N ranks on this node."""
    n = args.gpus
    if not args.launcher_selftest and not os.environ.get("HP_BENCH_SHARE_GPU"):
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} requested but only {have} HIP device(s) are visible; refusing to run on fewer",
                  file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def launcher_selftest(args) -> int:
    """The code around the timed region (rendezvous, barrier, max over ranks, one JSON line from rank 0) with gloo on CPU
    and a dummy step.  Exists so that `bench.py --gpus 2` can be exercised end to end where there is no GPU."""
    import torch.distributed as dist

    from hehub_amd import dist as hd

    world, rank, _ = hd.env_world()
    hd.init("gloo")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    step = lambda: time.sleep(0.002)
    for _ in range(args.warmup):
        step()
    hd.barrier(sync_device=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = hd.max_over_ranks(time.perf_counter() - t0)
    hd.barrier(sync_device=False)
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "value": world * args.steps / elapsed, "unit": "dummy-step/s",
                          "n_gpus": world, "rccl_ranks": ranks, "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none (launcher self-test: no engine work, no claim)",
                          "config": {"workload": "launcher self-test"}}))
    hd.finalize()
    return 0


# =====================================================================================================================
# inputs
# =====================================================================================================================
def rand_words(torch, shape, moduli, device, seed):
    """uniform words in [0, q_k) per limb (limb axis = -2), generated on the device"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(shape, dtype=torch.int64, device=device)
    for k, q in enumerate(moduli):
        out.select(-2, k).copy_(torch.randint(0, int(q), out.select(-2, k).shape, generator=g, device=device,
                                              dtype=torch.int64))
    return out


class Batch:
    """A batch of B items that repeats `period` distinct random classes (item i = class i mod period); period 0: all
    items distinct.  `base` keeps the classes (device), `full` is the batch the kernels see."""

    def __init__(self, torch, B, item_shape, moduli, device, seed, period):
        self.period = period if 0 < period < B else 0
        if self.period:
            self.base = rand_words(torch, (self.period,) + tuple(item_shape), moduli, device, seed)
            idx = torch.arange(B, device=device) % self.period
            self.full = self.base.index_select(0, idx).contiguous()
        else:
            self.full = rand_words(torch, (B,) + tuple(item_shape), moduli, device, seed)
            self.base = None
        self.B = B

    def classes(self, sample=(0,)):
        """(indices of the items that stand for all others, their host copies as uint64)"""
        import numpy as np

        if self.period:
            return list(range(self.period)), self.base.cpu().numpy().view(np.uint64)
        idx = sorted({i % self.B for i in sample})
        return idx, self.full[idx].cpu().numpy().view(np.uint64)


def compare_classes(torch, out, expected_host, period, idx):
    """every item of `out` against its class (period > 0) or the sampled items against theirs; returns (ok, compared)"""
    import numpy as np

    exp = torch.from_numpy(np.ascontiguousarray(expected_host).view(np.int64)).to(out.device)
    if period:
        ok = all(bool(torch.equal(out[c::period], exp[c].expand_as(out[c::period]))) for c in range(period))
        return ok, out.shape[0]
    ok = all(bool(torch.equal(out[i], exp[j])) for j, i in enumerate(idx))
    return ok, len(idx)


_CHECKER_BUILT = False


def build_checker_once(hd=None, rank=0):
    """compile the C restatement if its .so is missing or stale: rank 0 only (N ranks running make in one directory would
    race), the others wait at a barrier.  Building the checker is not using it."""
    global _CHECKER_BUILT
    if _CHECKER_BUILT:
        return
    from oracle.pyoracle import build

    if rank == 0:
        build(ref=False)
    if hd is not None:
        hd.barrier()
    _CHECKER_BUILT = True


def checker():
    from oracle.pyoracle import Oracle, have_ref

    build_checker_once()
    kind = "reference" if have_ref() else "port"
    return Oracle("ref" if kind == "reference" else "orc"), kind


# =====================================================================================================================
# CPU baseline (after the timed region; the checker, never the product)
# =====================================================================================================================
LOGN_OVERRIDE = 0   # set from --logn so that the CPU leg times the same ring degree


def cpu_baseline(workload, P, budget_s):
    """Time the CPU path on this host: the compiled reference when oracle/_ref travelled with the
    snapshot ("reference"), else the C restatement ("port").  Single thread, bounded sample."""
    from oracle.pyoracle import SplitMix

    lib, kind = checker()
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


def cpu_baseline_node(workload, procs, budget_s, logn_override):
    """The same sample as P independent processes at once: rates summed, P stated."""
    import multiprocessing as mp

    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if procs < 0:
        procs = max(1, min(visible, 64))
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.starmap(_cpu_baseline_worker, [(workload, budget_s, logn_override)] * procs)
    return dict(parts[0], value=sum(p["value"] for p in parts), cores=procs, cores_visible=visible,
                per_process_min=min(p["value"] for p in parts), per_process_max=max(p["value"] for p in parts),
                sample=f"{procs} concurrent single-threaded processes ({visible} hardware threads visible to this process), each: "
                       + parts[0]["sample"])


# =====================================================================================================================
# the other half of the metric: limb-transform and coefficient-wise rates (outside the hom-mult region)
# =====================================================================================================================
def timed_launches(torch, hd, eng, fn, family, steps, dev):
    """`steps` calls of fn between fences; returns (wall seconds max over ranks, launches, kernel ms from the library's
    HIP events on its stream)"""
    fn()
    hd.barrier()
    eng.prof_begin(family)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    launches, kern_ms = eng.prof_end()
    dt = hd.max_over_ranks(t1 - t0, device="cpu" if os.environ.get("HP_BENCH_SHARE_GPU") else dev)
    hd.barrier()
    return dt, launches, kern_ms


def rate_entry(units_per_launch, bytes_per_unit, steps, world, dt, launches, kern_ms):
    """units/s over the wall clock of the region (whole job) and the achieved GB/s of the kernel itself (events)"""
    e = {"per_s": units_per_launch * world * steps / dt,
         "wall_GBps_per_gpu": units_per_launch * bytes_per_unit * steps / dt / 1e9}
    if launches:
        gbps = units_per_launch * bytes_per_unit * steps / (kern_ms * 1e-3) / 1e9
        e.update({"avg_launch_ms": kern_ms / launches, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS})
    return e


def transform_rates(torch, hd, eng, P, args, world, dev, rank):
    """forward / inverse limb transforms at N = 4096 .. 32768 (north star: "NTT/INTT ... at N in {4096..32768}"), the C3
    ciphertext moduli (2^16 | q - 1 for all of them), 2.5 GiB in place per launch = the limbs of the C3 ciphertext batch (5120 at
    N = 32768, 20 per CU); one application of each transform checked against the checker on the periodic batch afterwards"""
    import numpy as np

    moduli = P.C3_Q
    L = len(moduli)
    out = {}
    lib = None if args.no_verify else checker()[0]
    # (key, log2 N, polynomials): by-N entries at the bytes of the C3 ciphertext batch (256 x 2 polynomials of 10 limbs at
    # N = 32768 = 2.5 GiB), plus "steady": as many limb transforms per launch as the digit-spread launch of the C3 step (25 600 =
    # 100 per CU; a launch's first and last rounds cost about one and a half items, which 20 rounds do not amortise)
    shapes = [(str(1 << logn), logn, (512 << 15) >> logn) for logn in (12, 13, 14, 15)] + [("steady_32768", 15, 2560)]
    for key, logn, B in shapes:
        n = 1 << logn
        xb = Batch(torch, B, (L, n), moduli, dev, 40 + logn + 100 * rank, 3)
        x = xb.full
        ent = {"N": n, "limbs_per_launch": B * L, "bytes_in_place": B * L * n * 8}
        for name, fam, fn in (("forward", "ntt", lambda: eng.ntt_(moduli, x)), ("inverse", "intt", lambda: eng.intt_(moduli, x))):
            dt, launches, kern_ms = timed_launches(torch, hd, eng, fn, fam, args.steps, dev)
            ent[name] = rate_entry(B * L, 16.0 * n, args.steps, world, dt, launches, kern_ms)
            ent[name]["unit"] = "limb-NTT/s"
        if lib is not None:
            idx, host = xb.classes()
            y = xb.base.index_select(0, torch.arange(B, device=dev) % xb.period).contiguous()   # fresh copy of the inputs
            eng.ntt_(moduli, y)
            fwd = np.stack([lib.poly_ntt(moduli, host[c]) for c in range(len(idx))])
            ok1, cnt = compare_classes(torch, y, fwd, xb.period, idx)
            eng.intt_(moduli, y)
            inv = np.stack([lib.poly_intt(moduli, fwd[c]) for c in range(len(idx))])
            ok2, _ = compare_classes(torch, y, inv, xb.period, idx)
            ent["verified"] = bool(ok1 and ok2)
            ent["verified_polynomials"] = cnt
        out[key] = ent
        del x, xb
    return out


def coeffwise_rates(torch, hd, eng, P, args, world, dev, rank):
    """RnsPolynomial operator* (hybrid Montgomery + Harvey product, rns.cpp:120-140) and operator+= (rns.cpp:58-87) at the
    C3 limb shape: 24*N algorithmic bytes per limb; 3 x 512 MiB touched per launch"""
    import numpy as np

    moduli = P.C3_Q
    L, n = len(moduli), 1 << P.C3_LOGN
    B = (512 << 20) // (8 * n * L)
    a = Batch(torch, B, (L, n), moduli, dev, 61 + 100 * rank, 3)
    b = Batch(torch, B, (L, n), moduli, dev, 62 + 100 * rank, 3)
    o = eng.empty((B, L, n))
    out = {"N": n, "limbs_per_launch": B * L}
    lib = None if args.no_verify else checker()[0]
    for name, fn, ref in (("mul", lambda: eng.poly_mul(moduli, a.full, b.full, out=o), "poly_mul"),
                          ("add", lambda: eng.poly_add(moduli, a.full, b.full, out=o), "poly_add")):
        dt, launches, kern_ms = timed_launches(torch, hd, eng, fn, "elem", args.steps, dev)
        out[name] = rate_entry(B * L, 24.0 * n, args.steps, world, dt, launches, kern_ms)
        out[name]["unit"] = "limb-op/s"
        if lib is not None:
            idx, ha = a.classes()
            _, hb = b.classes()
            exp = np.stack([getattr(lib, ref)(moduli, ha[c], hb[c]) for c in range(len(idx))])
            ok, cnt = compare_classes(torch, o, exp, a.period, idx)
            out[name]["verified"] = bool(ok)
            out[name]["verified_polynomials"] = cnt
    return out


def ckks_rates(torch, hd, eng, P, args, world, dev, rank):
    """ckks::mult + relinearize + rescale_inplace at the smaller ring degrees the north star names (N = 4096, 8192, 16384; the C3
    moduli chain, L = 10, batch 256 per GPU), timed like the headline and every output checked against the checker"""
    import numpy as np

    mext = P.C3_MODULI_EXT
    L, B = len(mext) - 1, P.C3_BATCH
    out = {}
    lib = None if args.no_verify else checker()[0]
    for logn in (12, 13, 14):
        n = 1 << logn
        b1 = Batch(torch, B, (2, L, n), mext[:L], dev, 300 + logn + 100 * rank, 3)
        b2 = Batch(torch, B, (2, L, n), mext[:L], dev, 400 + logn + 100 * rank, 3)
        key = rand_words(torch, (L, 2, L + 1, n), mext, dev, 7 + logn)
        res = eng.empty((B, 2, L - 1, n))
        dt, _, _ = timed_launches(torch, hd, eng, lambda: eng.ckks_mult(mext, b1.full, b2.full, key, out=res), "none", args.steps, dev)
        a_step = (5 * L * L + 36 * L) * 8 * n
        ent = {"N": n, "L": L, "batch_per_gpu": B, "per_s": B * world * args.steps / dt, "unit": "hom-mult/s",
               "A_step_frac_of_hbm_peak": B * args.steps / dt * a_step / 1e9 / HBM_PEAK_GBS}
        if lib is not None:
            idx, h1 = b1.classes()
            _, h2 = b2.classes()
            hk = key.cpu().numpy().view(np.uint64)
            exp = np.stack([lib.ckks_mult(mext, h1[c], h2[c], hk) for c in range(len(idx))])
            ok, cnt = compare_classes(torch, res, exp, b1.period, idx)
            ent["verified"] = bool(ok)
            ent["verified_outputs"] = cnt
        out[str(n)] = ent
    return out


# =====================================================================================================================
def main() -> int:
    global LOGN_OVERRIDE
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if args.launcher_selftest:
        return launcher_selftest(args)
    LOGN_OVERRIDE = args.logn
    import numpy as np
    import torch
    import torch.distributed as dist

    import params as P
    from hehub_amd.engine import Engine

    from hehub_amd import dist as hd

    world, rank, local = hd.env_world()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for the wrong job size",
              file=sys.stderr)
        return 2
    # TEST MODE (tests/test_bench_contract.py): the ranks of a multi-rank run share GPU 0 and rendezvous over gloo, so that the whole
    # N > 1 code path of this file runs on a one-GPU box.  The line says so; it is not a scaling number.
    share_gpu = bool(os.environ.get("HP_BENCH_SHARE_GPU")) and world > 1
    if share_gpu:
        local = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        print(f"bench.py: rank {rank} needs HIP device {local}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} "
              f"visible (no CPU fallback)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    if share_gpu:
        hd.init("gloo")
    else:
        hd.init("nccl", device=torch.device(f"cuda:{local}"))   # "nccl" is RCCL on ROCm; rendezvous + timing fences only
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1
    dev = f"cuda:{local}"
    cdev = "cpu" if share_gpu else dev     # where the tensors of the few collectives live
    eng = Engine(local)
    period = args.input_period
    verify = None          # callable -> (ok, compared, classes) run after the timed region
    extras = not args.roofline_only

    wl = args.workload
    scaling = "weak"
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
        xb = Batch(torch, B, (L, n), moduli, dev, 2 + rank, period)
        x = xb.full
        units_per_step = B * L
        step = (lambda: eng.intt_(moduli, x)) if inverse else (lambda: eng.ntt_(moduli, x))
        family = "intt" if inverse else "ntt"
        alg_bytes_per_step = 16.0 * n * B * L
        metric, unit = "limb_ntt_per_s", "limb-NTT/s"
        cfg = {"workload": f"{'C2' if (logn == 14 and L == 4) else 'C3-shape' if logn == 15 else 'custom'}: batched {'inverse' if inverse else 'forward'} negacyclic NTT, N={n}, {L} RNS limbs, batch={B} polynomials per GPU",
               "N": n, "limbs": L, "batch_per_gpu": B}

        def verify():
            # the timed buffer has been transformed in place steps+warmup times; check one fresh application of the same call
            lib, kind = checker()
            idx, host = xb.classes((0, B // 2, B - 1))
            y = (xb.base.index_select(0, torch.arange(B, device=dev) % xb.period).contiguous() if xb.period else
                 rand_words(torch, (B, L, n), moduli, dev, 2 + rank))
            (eng.intt_ if inverse else eng.ntt_)(moduli, y)
            exp = np.stack([(lib.poly_intt if inverse else lib.poly_ntt)(moduli, host[c]) for c in range(len(idx))])
            ok, cnt = compare_classes(torch, y, exp, xb.period, idx)
            return ok, cnt, len(idx), kind
    elif wl in ("mul", "add"):
        # coefficient-wise kernels at the C2 shape: RnsPolynomial operator* (hybrid Montgomery+Harvey product,
        # rns.cpp:120-140) / operator+= (rns.cpp:58-87); 24*N algorithmic bytes per limb
        logn, moduli = P.C2_LOGN, P.C2_MODULI
        B = args.batch or P.C2_BATCH
        n, L = 1 << logn, len(moduli)
        ab = Batch(torch, B, (L, n), moduli, dev, 21 + rank, period)
        bb = Batch(torch, B, (L, n), moduli, dev, 22 + rank, period)
        a, b = ab.full, bb.full
        out = eng.empty((B, L, n))
        units_per_step = B * L
        step = (lambda: eng.poly_mul(moduli, a, b, out=out)) if wl == "mul" else (lambda: eng.poly_add(moduli, a, b, out=out))
        family = "elem"
        alg_bytes_per_step = 24.0 * n * B * L
        metric, unit = f"limb_{wl}_per_s", "limb-op/s"
        cfg = {"workload": f"C2 shape: coefficient-wise modular {'multiply' if wl == 'mul' else 'add'}, N={n}, {L} RNS limbs, batch={B} polynomials per GPU",
               "N": n, "limbs": L, "batch_per_gpu": B}

        def verify():
            lib, kind = checker()
            idx, ha = ab.classes((0, B // 2, B - 1))
            _, hb = bb.classes((0, B // 2, B - 1))
            f = lib.poly_mul if wl == "mul" else lib.poly_add
            exp = np.stack([f(moduli, ha[c], hb[c]) for c in range(len(idx))])
            ok, cnt = compare_classes(torch, out, exp, ab.period, idx)
            return ok, cnt, len(idx), kind
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
        shared = wl == "ckks-limb"            # the limb-sharded mode works on the SAME batch on every rank
        b1 = Batch(torch, B, (2, L, n), mext[:L], dev, 3 + (0 if shared else rank), period)
        b2 = Batch(torch, B, (2, L, n), mext[:L], dev, 1003 + (0 if shared else rank), period)
        ct1, ct2 = b1.full, b2.full
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
        result = {"t": out}
        if wl == "ckks-limb":
            # latency mode (hehub_amd/sharded.py): the SAME small batch on every rank, cut by output modulus, with
            # the all-gather of the key-switch digits over RCCL; total work is fixed as N grows -> strong scaling
            from hehub_amd.sharded import Comm, ShardedMult

            comm, sm = Comm(), ShardedMult(eng, mext, world)
            bufs = sm.buffers(B, n)

            def step():
                result["t"] = sm.run(comm, ct1, ct2, key, bufs)
            metric, unit = "ckks_hom_mult_per_s", "hom-mult/s"
            name = "C3 shape, limb-sharded latency mode: ckks::mult + relinearize + rescale_inplace"
            scaling = "strong"
            units_per_step = B / world   # `value` multiplies by world below: the batch is shared, not replicated work
            check = lambda lib, c1, c2, k: lib.ckks_mult(mext, c1, c2, k)
        elif wl == "rotate":
            rot = eng.empty((B, 2, L, n))
            result["t"] = rot

            def step():
                result["t"] = eng.ckks_rotate(mext, ct1, key, 1)
            metric, unit = "ckks_rotation_per_s", "rotation/s"
            name = "C3 shape: ckks::rotate (gather + key switch + drop of the special prime)"
            a_limbs = 5 * L * L + 22 * L + 6
            check = lambda lib, c1, c2, k: lib.ckks_rotate(mext, c1, k, 1)
        elif wl == "ckks":
            step = lambda: eng.ckks_mult(mext, ct1, ct2, key, out=out)
            metric, unit = "ckks_hom_mult_per_s", "hom-mult/s"
            name = "C3: ckks::mult + relinearize + rescale_inplace"
            check = lambda lib, c1, c2, k: lib.ckks_mult(mext, c1, c2, k)
        else:
            step = lambda: eng.bgv_mult(mext, t, ct1, ct2, key, out=out)
            metric, unit = "bgv_hom_mult_per_s", "hom-mult/s"
            name = "C5 shape: bgv mult_low_level + relinearize + mod_switch_inplace"
            check = lambda lib, c1, c2, k: lib.bgv_mult(mext, t, c1, c2, k)
        family = "ntt"
        alg_bytes_per_step = 16.0 * n * fwd_per_ct * B
        cfg = {"workload": f"{name}, N={n}, L={L} moduli + special prime, batch={B} ciphertext pairs per GPU",
               "N": n, "L": L, "batch_per_gpu": B, "sub_batch": int(os.environ.get("HP_MULT_CHUNK", "0")) or B,
               "input_period": b1.period, "A_step_bytes_per_op": a_limbs * 8 * n}

        def verify():
            # the buffer the LAST timed step wrote, every ciphertext of it
            lib, kind = checker()
            sample = (0, 1, B // 2, B - 1)
            idx, h1 = b1.classes(sample)
            _, h2 = b2.classes(sample)
            hk = key.cpu().numpy().view(np.uint64)
            exp = np.stack([check(lib, h1[c], h2[c], hk) for c in range(len(idx))])
            res = result["t"]
            if res.shape[0] != B:
                return False, 0, len(idx), kind
            ok, cnt = compare_classes(torch, res, exp, b1.period, idx)
            return ok, cnt, len(idx), kind

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
    elapsed = hd.max_over_ranks(t1 - t0, device=cdev)
    hd.barrier()

    value = units_per_step * world * args.steps / elapsed
    res = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": cfg,
    }
    if share_gpu:
        res["data"] = "synthetic (TEST MODE: the ranks share ONE GPU over gloo; exercises the multi-rank path, not a scaling number)"
        res["backend"] = "gloo"
    # ---- output check on the timed buffers (every rank checks its own; the verdict is the AND over ranks) ----------
    failed = False
    if extras:
        build_checker_once(hd, rank)
    if extras and not args.no_verify and verify is not None:
        try:
            ok, compared, classes, kind = verify()
        except Exception as e:   # a missing checker must not look like a pass
            ok, compared, classes, kind = False, 0, 0, f"error: {e!r}"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        if dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["verified"] = bool(flag.item())
        res["verify"] = {"outputs_compared_per_gpu": compared, "checker_evaluations": classes, "checker": kind,
                         "what": "raw u64 words of the buffers the timed region wrote (in-place transforms: one fresh call on "
                                 "the same inputs), bit for bit"}
        failed = not res["verified"]
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
        # HBM traffic per launch and VALUBusy from the committed PMC measurement of this kernel shape (rocprofv3 --pmc
        # FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, tools/prof_pmc.sh); null when no measurement exists
        try:
            with open(TRAFFIC_FILE) as f:
                trs = json.load(f)
            tr = None
            if family == "ntt":   # the digit-spread launch has its own measurement where one exists
                tr = (trs.get(f"k_ntt_fwd_logn{logn}_spread") if wl in ("ckks", "bgv", "rotate") else None) or trs.get(f"k_ntt_fwd_logn{logn}")
            elif family == "intt":
                tr = trs.get(f"k_ntt_inv_logn{logn}")
            if tr:
                limbs_per_launch = bytes_per_launch / (16.0 * n)
                res["roofline"]["traffic"] = tr["bytes_per_limb"] * limbs_per_launch
                res["roofline"]["traffic_source"] = ("rocprofv3 PMC per-limb measurement x limbs per launch "
                                                     f"(profiles/traffic.json: {tr.get('source', 'see _comment')})")
                if "valu_busy" in tr:
                    res["roofline"]["valu_busy"] = tr["valu_busy"]
        except (OSError, ValueError, KeyError):
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
    if wl == "ckks" and extras and not args.no_rates and not args.logn and not args.batch:
        # BASELINE.json's metric names both rates ("NTT/s and CKKS hom-mult/s ... N=32768") and the north star the
        # coefficient-wise kernels: the default line carries them too (timed after the hom-mult region, same fences)
        by_n = transform_rates(torch, hd, eng, P, args, world, dev, rank)
        top = by_n[str(1 << P.C3_LOGN)]
        steady = by_n.pop("steady_32768")
        res["ntt"] = {"N": top["N"], "limbs_per_launch": top["limbs_per_launch"],
                      "forward_limb_ntt_per_s": top["forward"]["per_s"], "inverse_limb_ntt_per_s": top["inverse"]["per_s"],
                      "forward": top["forward"], "inverse": top["inverse"], "verified": top.get("verified"),
                      "steady_state": steady, "by_N": by_n}
        res["coeffwise"] = coeffwise_rates(torch, hd, eng, P, args, world, dev, rank)
        res["ckks_by_N"] = ckks_rates(torch, hd, eng, P, args, world, dev, rank)
        res["ckks_by_N"][str(n)] = {"N": n, "L": L, "batch_per_gpu": B, "per_s": value, "unit": "hom-mult/s",
                                    "A_step_frac_of_hbm_peak": res["pipeline_roofline"]["frac_of_hbm_peak"],
                                    "verified": res.get("verified"), "verified_outputs": B}
        for sect in list(by_n.values()) + list(res["ckks_by_N"].values()) + [steady, res["coeffwise"]["mul"], res["coeffwise"]["add"]]:
            if sect.get("verified") is False:
                failed = True
    if rank == 0:
        if world == 1 and extras and not args.no_cpu_baseline:
            cwl = "ckks" if wl in ("ckks-limb", "ckks-hks") else wl
            try:
                res["cpu_baseline"] = cpu_baseline(cwl, P, args.cpu_seconds)
            except Exception as e:  # the checker is optional infrastructure; the GPU number stands on its own
                res["cpu_baseline"] = {"error": repr(e)}
            if args.cpu_procs != 0:
                try:
                    res["cpu_baseline_node"] = cpu_baseline_node(cwl, args.cpu_procs, args.cpu_node_seconds, args.logn)
                except Exception as e:
                    res["cpu_baseline_node"] = {"error": repr(e)}
        print(json.dumps(res))
    hd.finalize()
    eng.close()
    if failed:
        print("bench.py: OUTPUT CHECK FAILED -- the numbers above are void", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
