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
(SURVEY.md section 8e), so per-GPU work is fixed: weak scaling.  Rank 0 prints, as its LAST stdout line, ONE compact JSON line
(< 8 KB: the contract fields, `roofline`, `cpu_baseline`, `summary` = one number per section); the detailed sections named below go
to `#section <name> <json>` lines before it and to bench_sections.json (benchkit/line.py; `tools/benchline.py` merges them back):

  value / ms_per_step   whole-job throughput of the timed region (barrier + synchronize on both sides, max over ranks)
  roofline              dominant kernel (digit-spread k_ntt_fwd launch), HIP events recorded by the library on its stream
  verified              every output of the timed buffers compared with the CPU checker AFTER the timed region: the batch
                        is periodic (ciphertext i = class i mod 3 of three distinct random inputs, --input-period), so
                        three checker evaluations cover all of them; a mismatch makes the run fail (exit code 1)
  ntt / coeffwise       (default ckks line) the other half of BASELINE's metric: forward / inverse limb-NTT/s at N = 32768
                        and across N = 4096..32768, coefficient-wise multiply / add, each with achieved GB/s and the
                        fraction of HBM peak; timed outside the hom-mult region
  ckks_by_N             (default ckks line) hom-mult/s at N = 4096 .. 32768 (same chain, batch 256), each verified
  c2                    (default ckks line) BASELINE config 2 exactly: N = 16384, four 50-bit primes, 1024 polynomials;
                        forward and inverse passes timed separately, all 4096 limbs verified
  bgv                   (default ckks line) BASELINE config 5 per-GPU shape: N = 8192, L = 6, t = 65537, batch 512:
                        hom-mult/s, A_step fraction, the dominant kernel's roofline, all 512 outputs verified, CPU sample
  hbm_copy_ceiling_GBps (default ckks line) the measured device-to-device stream rate beside the 8 TB/s spec peak
  object_api            (default ckks line) the same C3 hom-mult through hehub's OBJECT interface (hehub_amd/host/hehub.hpp): 256
                        independent ckks::mult + rescale_inplace as single calls, as ONE batched call (hehub_amd_ext.hpp), and
                        independent chains over 1 / 8 lanes; examples/independent_mults as a child process, digests compared
                        object_api.matvec: the diagonal loop of hehub's matrix_vector_mul_short (linear_algebra.h:104-136) at the C3
                        shape, width 16 (30 rotations of one vector under 30 keys): eager / deferred / batched form, hehub's digest,
                        hehub itself on this host's CPU beside it (examples/diag_matvec, oracle/_ref/ref_matvec_cpu)
                        object_api.reference_benchmark: hehub's OWN benchmark (bench/benchmarks.cpp:21-37: ckks::rotate, one ciphertext
                        per call, N = 2^12 .. 2^15 with create_params' modulus chains) on synthetic words: ms per rotation with a look
                        after every call / back to back / recorded, hehub's digests, hehub itself on this host's CPU beside it
                        (examples/rotate_bench, oracle/_ref/ref_rotbench_cpu)
  cpu_baseline          the compiled reference (or the C restatement) on ONE core of this host; cpu_baseline_node: the
                        same as P independent processes (P = what affinity mask and CPU quota allow, stated)
  step                  (pipelines) a pass after the timed region with HIP events around EVERY launch: per kernel family launches,
                        ms, its share of A_step and the HBM fraction that makes; `step_traffic`: the measured HBM bytes of one
                        step (sum of 2 x FETCH_SIZE + WRITE_SIZE over every kernel, committed rocprofv3 passes of this command)
                        beside A_step / A_min -- a pipeline_roofline above 1 is cross-stage fusion, not a measurement error
  --roofline-only       drops everything after the timed region (for clean rocprofv3 summaries of the default command)

The workloads, inputs, fences and sections live in benchkit/ (no checker there).  The oracle / compiled reference is used
only here, only after the timed region, as the checker and as the CPU baseline.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

# the host driver of these boxes only supports dmabuf IPC: RCCL needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# pageable numpy <-> device copies (inputs, the verify pass) through the runtime's staging buffers, not by pinning numpy's pages in
# place: tests/conftest.py has the why
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from benchkit import host as bhost                      # noqa: E402
from benchkit.launcher import launcher_selftest, self_launch   # noqa: E402
from benchkit.timing import HBM_PEAK_GBS                # noqa: E402
from benchkit.workloads import NAMES as WORKLOADS       # noqa: E402


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
    ap.add_argument("--no-object-api", action="store_true", help="skip the object_api section (examples/independent_mults as a child process)")
    ap.add_argument("--no-rates", action="store_true", help="ckks workload: skip the extra sections (ntt, coeffwise, c2, bgv, ...)")
    ap.add_argument("--hks-alpha", type=int, default=2, help="ckks-hks: ciphertext moduli per key-switch digit")
    ap.add_argument("--hks-k", type=int, default=2, help="ckks-hks: number of (50-bit) special primes")
    ap.add_argument("--limb-transport", default=None, choices=["p2p", "allgather"],
                    help="ckks-limb: exchange of the key-switch digits as batched peer-to-peer sends (default) or as ONE "
                         "all_gather collective (ncclAllGather on RCCL)")
    ap.add_argument("--no-force-dist", action="store_true",
                    help="ONE rank: do not create the torch.distributed process group.  By default a one-rank run creates it too "
                         "(backend nccl = RCCL, world size 1), so that the barrier, the MAX / MIN all-reduces on device tensors and "
                         "the communicator set-up of an N-GPU job run on a one-GPU box; a failure there is recorded in the line "
                         "(`rccl.error`) and the run goes on without a group")
    ap.add_argument("--parity-level", default=None, choices=["A", "B"],
                    help="parity level of the scheme-level pipelines of the timed workload (include/hehub_amd.h: hp_ctx_set_parity_level): "
                         "B (default) = hehub's raw lazy words; A = canonical residues through the FP64 transforms, verified as "
                         "reduce_strict of the checker's words.  The default line is always level B and reports level A in its own "
                         "`level_a` section; this switch exists for profiling one level alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU budget of the cpu_baseline sample")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="cpu_baseline_node: this many independent processes at once (the reference has process-global "
                         "unsynchronised caches, so processes, not threads); -1 = what the affinity mask, the cpuset and the "
                         "cgroup CPU quota of this process allow (at most 64), 0 = skip")
    ap.add_argument("--cpu-node-seconds", type=float, default=5.0, help="CPU budget per process of cpu_baseline_node")
    ap.add_argument("--cpu-section-seconds", type=float, default=1.5, help="CPU budget of the samples inside the c2 / bgv sections")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="exercise only the launcher / rendezvous / timing fences with the gloo backend and a dummy step "
                         "(no engine, no GPU); the line says so and carries no throughput claim")
    return ap.parse_args(argv)


# =====================================================================================================================
# the CPU checker (after the timed region only; never the product)
# =====================================================================================================================
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


def _loop(f, budget_s):
    """warm call, then calls until the budget is spent; returns (seconds per call, calls)"""
    f()
    iters, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        f(); iters += 1
    return (time.perf_counter() - t0) / iters, iters


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
        per, iters = _loop(f, budget_s)
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
            per, iters = _loop(lambda: f(logn, q, x), budget_s)
        return {"value": 1.0 / per, "unit": "limb-NTT/s", "cores": 1, "kind": kind,
                "sample": f"{iters} {'inverse' if inv else 'forward'} NTTs of one limb, N={1 << logn}, q={q}, single thread, tables warm"}
    if workload == "encdec":
        logn, moduli = P.C3_LOGN, P.C3_Q
        noise, c1, pt, sk = P.edge_case(rng, logn, moduli)
        per, iters = _loop(lambda: lib.rlwe_decrypt_core(moduli, lib.rlwe_encrypt_core(moduli, noise, c1, pt, sk), sk), budget_s)
        return {"value": 1.0 / per, "unit": "ciphertext/s", "cores": 1, "kind": kind,
                "sample": f"{iters} x (encrypt_core on given samples + decrypt_core) of one ciphertext, N={1 << logn}, L={len(moduli)}, single thread, tables warm"}
    if workload == "rotate":   # the reference's own benchmark workload (bench/benchmarks.cpp:21-37) at the C3 shape
        logn, mext = LOGN_OVERRIDE or P.C3_LOGN, P.C3_MODULI_EXT
        n, L = 1 << logn, len(mext) - 1
        ct = rng.poly((2, L, n), mext[:L])
        key = rng.poly((L, 2, L + 1, n), mext)
        per, iters = _loop(lambda: lib.ckks_rotate(mext, ct, key, 1), budget_s)
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
        per, iters = _loop(f, budget_s)
    name = "ckks::mult+relinearize+rescale_inplace" if workload == "ckks" else "bgv mult+relinearize+mod_switch"
    return {"value": 1.0 / per, "unit": "hom-mult/s", "cores": 1, "kind": kind,
            "sample": f"{iters} x {name} on one ciphertext pair, N={n}, L={L}, single thread, tables warm"}


def _cpu_baseline_worker(workload, budget_s, logn_override=0):
    import params as P

    global LOGN_OVERRIDE
    LOGN_OVERRIDE = logn_override
    return cpu_baseline(workload, P, budget_s)


def cpu_baseline_node(workload, procs, budget_s, logn_override, single=None):
    """The same sample as P independent processes at once: rates summed, P stated.  P = the cores this process
    can actually keep busy: min(affinity mask, cpuset, cgroup CPU quota, 64).  `single` = the one-process rate measured just
    before: when the slowest of the P processes runs below 0.8 x of it the line says so (`host_limited`) -- memory bandwidth
    or a quota this process cannot see, not the algorithm."""
    import multiprocessing as mp

    auto, facts = bhost.usable_cores(64)
    if procs < 0:
        procs = auto
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.starmap(_cpu_baseline_worker, [(workload, budget_s, logn_override)] * procs)
    pmin = min(p["value"] for p in parts)
    out = dict(parts[0], value=sum(p["value"] for p in parts), cores=procs, per_process_min=pmin,
               per_process_max=max(p["value"] for p in parts), **facts,
               sample=f"{procs} concurrent single-threaded processes ({facts['cores_visible']} hardware threads visible, CPU quota "
                      f"{facts['cpu_quota_cores'] or 'none'}), each: " + parts[0]["sample"])
    if single:
        out["parallel_efficiency"] = out["value"] / (procs * single)
        out["host_limited"] = bool(pmin < 0.8 * single)
    return out


# =====================================================================================================================
def extra_sections(run, res, wl, value, lib, cpu_budget):
    """the default line's sections beyond the headline (benchkit/sections.py); returns True when an output check failed"""
    from benchkit import sections as S

    P = run.P
    sec = run.chip.section
    with sec("ntt"):
        by_n = S.transform_rates(run, lib)
    top = by_n[str(1 << P.C3_LOGN)]
    steady = by_n.pop("steady_32768")
    res["ntt"] = {"N": top["N"], "limbs_per_launch": top["limbs_per_launch"],
                  "forward_limb_ntt_per_s": top["forward"]["per_s"], "inverse_limb_ntt_per_s": top["inverse"]["per_s"],
                  "forward": top["forward"], "inverse": top["inverse"], "verified": top.get("verified"),
                  "steady_state": steady, "by_N": by_n}
    with sec("c2"):
        res["c2"] = S.c2_section(run, lib)
    with sec("coeffwise"):
        res["coeffwise"] = S.coeffwise_rates(run, lib)
    with sec("ckks_by_N"):
        res["ckks_by_N"] = S.ckks_rates(run, lib)
    res["ckks_by_N"][str(wl.n)] = {"N": wl.n, "L": wl.L, "batch_per_gpu": wl.B, "per_s": value, "unit": "hom-mult/s",
                                   "A_step_frac_of_hbm_peak": res["pipeline_roofline"]["frac_of_hbm_peak"],
                                   "verified": res.get("verified"), "verified_outputs": wl.B}
    with sec("bgv"):
        res["bgv"] = S.bgv_section(run, lib)
    with sec("level_a"):
        res["level_a"] = S.level_a_section(run, lib, wl)
    res["level_a"]["ckks"]["speedup_vs_level_b"] = res["level_a"]["ckks"]["per_s"] / value
    res["level_a"]["bgv"]["speedup_vs_level_b"] = res["level_a"]["bgv"]["per_s"] / res["bgv"]["per_s"]
    with sec("hbm_copy"):
        copy = S.hbm_copy_ceiling(run)
    res["hbm_copy_ceiling_GBps"] = copy["hbm_copy_ceiling_GBps"]
    res["hbm_stream_ceiling_GBps"] = copy["hbm_stream_ceiling_GBps"]
    res["hbm_copy"] = copy
    if run.rank == 0 and run.world == 1 and not run.args.no_object_api:
        with sec("object_api"):
            try:
                res["object_api"] = S.object_api_section(run)
            except Exception as e:   # (a missing host compiler must not void the headline; the section says so)
                res["object_api"] = {"error": repr(e)[:300], "verified": None}
    if run.rank == 0 and run.world == 1 and cpu_budget > 0 and lib is not None:
        # the CPU path beside the two BASELINE configs that are not the headline (bounded samples, one core)
        # (... and beside the N = 32768 limb-transform rates, the other half of BASELINE's metric)
        for sect, names in ((res["ntt"], (("forward", "ntt15"), ("inverse", "intt15"))),
                            (res["c2"], (("forward", "ntt"), ("inverse", "intt"))), (res["bgv"], ((None, "bgv"),))):
            for key, name in names:
                try:
                    cb = cpu_baseline(name, P, cpu_budget)
                except Exception as e:
                    cb = {"error": repr(e)}
                (sect if key is None else sect[key])["cpu_baseline"] = cb
    checks = list(by_n.values()) + list(res["ckks_by_N"].values()) + [steady, res["coeffwise"]["mul"], res["coeffwise"]["add"],
                                                                        res["c2"], res["bgv"], res["level_a"]["ckks"],
                                                                        res["level_a"]["bgv"]] + [v for v in res["level_a"]["ntt"].values()
                                                                                                  if isinstance(v, dict)]
    checks.append(res.get("object_api", {}))
    bad = any(s.get("verified") is False for s in checks) or not copy["engine_copy_verified"]
    res["all_sections_verified"] = not bad
    return bad


def main() -> int:
    global LOGN_OVERRIDE
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, __file__)
    if args.launcher_selftest:
        return launcher_selftest(args)
    LOGN_OVERRIDE = args.logn
    import torch
    import torch.distributed as dist

    import params as P
    from benchkit import workloads
    from benchkit.timing import Run, roofline_entry
    from hehub_amd import dist as hd
    from hehub_amd.engine import Engine

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
    cpus_all = os.sched_getaffinity(0)
    placement = bhost.bind_to_gpu_numa(local)   # this rank's process on the socket of its GPU (two sockets x four GPUs on an MI355X node)
    rccl_info = {"initialised": False}
    if share_gpu:
        hd.init("gloo")
    else:
        t_init = time.perf_counter()
        try:   # "nccl" is RCCL on ROCm; rendezvous + timing fences only
            hd.init("nccl", device=torch.device(f"cuda:{local}"), force=(world == 1 and not args.no_force_dist))
            if dist.is_initialized():
                hd.barrier()               # the first collective creates the communicator
                rccl_info = {"initialised": True, "backend": dist.get_backend(), "init_s": round(time.perf_counter() - t_init, 3),
                             "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        except Exception as e:
            if world > 1:
                raise
            rccl_info = {"initialised": False, "error": repr(e)[:300]}
            if dist.is_initialized():   # the group exists but its first collective failed: later fences must not use it
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
    # dist_ranks: size of the process group whatever its backend (1 = none); rccl_ranks: ranks of an initialised RCCL communicator
    dist_ranks = dist.get_world_size() if dist.is_initialized() else 1
    rccl_ranks = dist_ranks if (dist.is_initialized() and dist.get_backend() == "nccl") else 0
    dev = f"cuda:{local}"
    run = Run(torch=torch, hd=hd, eng=Engine(local), P=P, args=args, world=world, rank=rank, dev=dev,
              cdev="cpu" if share_gpu else dev)
    from benchkit.chip import ChipSampler, cpu_model

    chip = ChipSampler(torch, local)
    run.chip = chip
    extras = not args.roofline_only
    env_level = os.environ.get("HP_PARITY_LEVEL", "")   # (parsed as hp_ctx_create does: an empty value is level B)
    level = args.parity_level or ("A" if env_level and env_level[0] in "Aa1" else "B")
    run.eng.set_parity_level(level)
    wl = workloads.make(run, args.workload)

    # ---- the timed region --------------------------------------------------------------------------------------------
    for _ in range(args.warmup):
        wl.step()
    hd.barrier()                       # dist.barrier() + torch.cuda.synchronize()
    run.eng.prof_begin(wl.family)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    chip.sections["timed_region"] = chip.summarize(t0, t1)
    launches, kern_ms = run.eng.prof_end()
    elapsed = hd.max_over_ranks(t1 - t0, device=run.cdev)
    hd.barrier()

    value = wl.units_per_step * world * args.steps / elapsed
    res = {
        "metric": wl.metric, "value": value, "unit": wl.unit, "n_gpus": world, "dist_ranks": dist_ranks, "rccl_ranks": rccl_ranks, "rccl": rccl_info,
        "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": wl.scaling,
        "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": wl.cfg, "parity_level": level,
    }
    if share_gpu:
        res["data"] = "synthetic (TEST MODE: the ranks share ONE GPU over gloo; exercises the multi-rank path, not a scaling number)"
        res["backend"] = "gloo"
    # ---- output check on the timed buffers (every rank checks its own; the verdict is the AND over ranks) ----------
    failed = False
    lib = kind = None
    if extras:
        build_checker_once(hd, rank)
    if extras and not args.no_verify:
        try:
            lib, kind = checker()
        except Exception as e:   # a missing checker must not look like a pass
            kind = f"error: {e!r}"
    if extras and not args.no_verify and wl.verifiable:
        try:
            ok, compared, classes = wl.verify(lib, strict=True) if (level == "A" and isinstance(wl, workloads.Scheme)) else wl.verify(lib)
        except Exception as e:
            ok, compared, classes, kind = False, 0, 0, f"error: {e!r}"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=run.cdev)
        if dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res["verified"] = bool(flag.item())
        res["verify"] = {"outputs_compared_per_gpu": compared, "checker_evaluations": classes, "checker": kind,
                         "what": "raw u64 words of the buffers the timed region wrote (in-place transforms: one fresh call on "
                                 "the same inputs), bit for bit"}
        failed = not res["verified"]
    if launches:
        res["roofline"] = roofline_entry(wl.family, wl.alg_bytes_per_step, args.steps, launches, kern_ms, elapsed, wl.logn, wl.spread,
                                         level=level, sclk_mhz=chip.sections["timed_region"].get("sclk_MHz"))
    if isinstance(wl, workloads.Scheme) and wl.name in ("ckks", "bgv", "rotate") and extras:
        from benchkit.timing import step_kernels

        res["step"] = step_kernels(run, wl, level)   # every launch of one step + the measured HBM bytes of a step (after the timed region)
    if wl.a_limbs is not None:
        res["pipeline_roofline"] = wl.pipeline_roofline(value / world, HBM_PEAK_GBS)
    if args.workload == "ckks" and extras and not args.no_rates and not args.logn and not args.batch and level == "B":
        # BASELINE.json's metric names both rates ("NTT/s and CKKS hom-mult/s ... N=32768"), the north star the coefficient-wise
        # kernels, and configs 2 and 5 their own shapes: the default line carries them all (timed after the hom-mult region)
        if extra_sections(run, res, wl, value, lib, 0.0 if args.no_cpu_baseline else args.cpu_section_seconds):
            failed = True
    # shader clock and socket power of this rank's GPU while each section ran (amdgpu hwmon: freq1_input, power1_input)
    res["chip"] = dict(chip.sections, source="amdgpu hwmon sysfs (sclk freq1_input, socket power1_input), sampled every 4 ms by a thread of this process"
                       if chip.dir else "no amdgpu hwmon files visible: not sampled")
    res["cpu_model"] = cpu_model()
    res["placement"] = placement
    chip.close()
    if rank == 0:
        os.sched_setaffinity(0, cpus_all)   # the CPU baselines below use whatever cores the host gives, as before
        if world == 1 and extras and not args.no_cpu_baseline:
            cwl = wl.cpu_name
            try:
                res["cpu_baseline"] = cpu_baseline(cwl, P, args.cpu_seconds)
            except Exception as e:  # the checker is optional infrastructure; the GPU number stands on its own
                res["cpu_baseline"] = {"error": repr(e)}
            if args.cpu_procs != 0:
                try:
                    res["cpu_baseline_node"] = cpu_baseline_node(cwl, args.cpu_procs, args.cpu_node_seconds, args.logn,
                                                                 res["cpu_baseline"].get("value"))
                except Exception as e:
                    res["cpu_baseline_node"] = {"error": repr(e)}
        from benchkit.line import emit

        emit(res, sys.stdout)   # the sections (`#section <name> <json>` lines + bench_sections.json), then the compact contract line LAST
    hd.finalize()
    run.eng.close()
    if failed:
        print("bench.py: OUTPUT CHECK FAILED -- the numbers above are void", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
