"""The extra sections of the default line (timed AFTER the headline region, same fences): the other half of BASELINE's
metric (limb transforms per second), the coefficient-wise kernels, hom-mult/s at the smaller ring degrees, BASELINE's
config 2 (C2) and config 5 (C5, BGV) at their exact shapes, and the measured HBM stream ceiling.  `lib` is the CPU
checker handed in by bench.py (None: no output checks)."""
from __future__ import annotations

from .inputs import Batch, compare_classes, rand_words
from .timing import HBM_PEAK_GBS, Run, rate_entry, roofline_entry, step_kernels, timed_launches
from .workloads import Scheme, Transform


def _check_transforms(run: Run, lib, moduli, xb):
    """one forward and one inverse application on a fresh copy of the periodic batch, every polynomial compared"""
    import numpy as np

    idx, host = xb.classes()
    y = xb.fresh()
    run.eng.ntt_(moduli, y)
    fwd = np.stack([lib.poly_ntt(moduli, host[c]) for c in range(len(idx))])
    ok1, cnt = compare_classes(run.torch, y, fwd, xb.period, idx)
    run.eng.intt_(moduli, y)
    inv = np.stack([lib.poly_intt(moduli, fwd[c]) for c in range(len(idx))])
    ok2, _ = compare_classes(run.torch, y, inv, xb.period, idx)
    return bool(ok1 and ok2), cnt


def transform_rates(run: Run, lib):
    """forward / inverse limb transforms at N = 4096 .. 32768 (north star: "NTT/INTT ... at N in {4096..32768}"), the C3
    ciphertext moduli (2^16 | q - 1 for all of them), 2.5 GiB in place per launch = the limbs of the C3 ciphertext batch (5120 at
    N = 32768, 20 per CU); one application of each transform checked against the checker on the periodic batch afterwards"""
    P, eng, steps = run.P, run.eng, run.args.steps
    moduli = P.C3_Q
    L = len(moduli)
    out = {}
    # (key, log2 N, polynomials): by-N entries at the bytes of the C3 ciphertext batch (256 x 2 polynomials of 10 limbs at
    # N = 32768 = 2.5 GiB), plus "steady": as many limb transforms per launch as the digit-spread launch of the C3 step (25 600 =
    # 100 per CU; a launch's first and last rounds cost about one and a half items, which 20 rounds do not amortise)
    shapes = [(str(1 << logn), logn, (512 << 15) >> logn) for logn in (12, 13, 14, 15)] + [("steady_32768", 15, 2560)]
    for key, logn, B in shapes:
        n = 1 << logn
        xb = Batch(run.torch, B, (L, n), moduli, run.dev, 40 + logn + 100 * run.rank, 3)
        x = xb.full
        ent = {"N": n, "limbs_per_launch": B * L, "bytes_in_place": B * L * n * 8}
        for name, fam, fn in (("forward", "ntt", lambda: eng.ntt_(moduli, x)), ("inverse", "intt", lambda: eng.intt_(moduli, x))):
            dt, launches, kern_ms = timed_launches(run, fn, fam, steps)
            ent[name] = rate_entry(B * L, 16.0 * n, steps, run.world, dt, launches, kern_ms)
            ent[name]["unit"] = "limb-NTT/s"
        if lib is not None:
            ent["verified"], ent["verified_polynomials"] = _check_transforms(run, lib, moduli, xb)
        out[key] = ent
        del x, xb
    return out


def c2_section(run: Run, lib):
    """BASELINE config 2 exactly (bench/ntt_bm.cpp:9-26 is the reference's set-up): N = 16384, the first four
    50-bit list primes, 1024 polynomials x 4 limbs = 512 MiB in place, forward and inverse passes timed separately, every one
    of the 4096 limbs compared with the checker (periodic batch, fresh application)"""
    P, eng, steps = run.P, run.eng, max(run.args.steps, 20)
    wl = Transform(run, "ntt", 0, 0, 3, seed=2)
    moduli, x, n = wl.moduli, wl.x, wl.n
    ent = {"workload": wl.cfg["workload"], "N": n, "limbs": wl.L, "batch_per_gpu": wl.B, "limbs_per_launch": wl.B * wl.L,
           "bytes_in_place": wl.B * wl.L * n * 8, "moduli": [int(q) for q in moduli], "steps": steps}
    for name, fam, fn in (("forward", "ntt", lambda: eng.ntt_(moduli, x)), ("inverse", "intt", lambda: eng.intt_(moduli, x))):
        dt, launches, kern_ms = timed_launches(run, fn, fam, steps, warm=3)
        e = rate_entry(wl.B * wl.L, 16.0 * n, steps, run.world, dt, launches, kern_ms)
        e["unit"] = "limb-NTT/s"
        e["ms_per_pass"] = 1e3 * dt / steps
        if launches:
            e["roofline"] = roofline_entry(fam, 16.0 * n * wl.B * wl.L, steps, launches, kern_ms, dt, wl.logn, False, sclk_mhz=run.sclk())
        ent[name] = e
    if lib is not None:
        ent["verified"], polys = _check_transforms(run, lib, moduli, wl.xb)
        ent["verified_limbs"] = polys * wl.L
    return ent


def coeffwise_rates(run: Run, lib):
    """RnsPolynomial operator* (hybrid Montgomery + Harvey product, rns.cpp:120-140) and operator+= (rns.cpp:58-87) at the
    C3 limb shape: 24*N algorithmic bytes per limb; 3 x 512 MiB touched per launch"""
    import numpy as np

    P, eng, steps = run.P, run.eng, run.args.steps
    moduli = P.C3_Q
    L, n = len(moduli), 1 << P.C3_LOGN
    B = (512 << 20) // (8 * n * L)
    a = Batch(run.torch, B, (L, n), moduli, run.dev, 61 + 100 * run.rank, 3)
    b = Batch(run.torch, B, (L, n), moduli, run.dev, 62 + 100 * run.rank, 3)
    o = eng.empty((B, L, n))
    out = {"N": n, "limbs_per_launch": B * L}
    for name, fn, ref in (("mul", lambda: eng.poly_mul(moduli, a.full, b.full, out=o), "poly_mul"),
                          ("add", lambda: eng.poly_add(moduli, a.full, b.full, out=o), "poly_add")):
        dt, launches, kern_ms = timed_launches(run, fn, "elem", steps)
        out[name] = rate_entry(B * L, 24.0 * n, steps, run.world, dt, launches, kern_ms)
        out[name]["unit"] = "limb-op/s"
        if lib is not None:
            idx, ha = a.classes()
            _, hb = b.classes()
            exp = np.stack([getattr(lib, ref)(moduli, ha[c], hb[c]) for c in range(len(idx))])
            ok, cnt = compare_classes(run.torch, o, exp, a.period, idx)
            out[name]["verified"] = bool(ok)
            out[name]["verified_polynomials"] = cnt
    return out


def ckks_rates(run: Run, lib):
    """ckks::mult + relinearize + rescale_inplace at the smaller ring degrees the north star names (N = 4096, 8192, 16384; the C3
    moduli chain, L = 10, batch 256 per GPU), timed like the headline and every output checked against the checker"""
    out = {}
    for logn in (12, 13, 14):
        wl = Scheme(run, "ckks", 0, logn, 3, seed=300 + logn)
        dt, _, _ = timed_launches(run, wl.step, "none", run.args.steps)
        per_gpu = wl.B * run.args.steps / dt
        ent = {"N": wl.n, "L": wl.L, "batch_per_gpu": wl.B, "per_s": per_gpu * run.world, "unit": "hom-mult/s",
               "A_step_frac_of_hbm_peak": wl.pipeline_roofline(per_gpu, HBM_PEAK_GBS)["frac_of_hbm_peak"]}
        if lib is not None:
            ok, cnt, _ = wl.verify(lib)
            ent["verified"] = bool(ok)
            ent["verified_outputs"] = cnt
        out[str(wl.n)] = ent
        del wl
    return out


def bgv_section(run: Run, lib):
    """BASELINE config 5 at its per-GPU shape (bgv/arith.cpp:59-79 + mod_switch.cpp:13-78; N = 8192, L = 6, t = 65537,
    4096 / 8 = 512 ciphertext pairs per GPU), timed like the headline; the dominant kernel's roofline entry (digit-spread
    k_ntt_fwd launch) from the library's events; every one of the 512 outputs compared with the checker"""
    steps = max(run.args.steps, 20)
    wl = Scheme(run, "bgv", 0, 0, 3, seed=5)
    dt, launches, kern_ms = timed_launches(run, wl.step, wl.family, steps, warm=3)
    per_gpu = wl.B * steps / dt
    ent = {"workload": wl.cfg["workload"], "N": wl.n, "L": wl.L, "plain_modulus": wl.t, "batch_per_gpu": wl.B,
           "per_s": per_gpu * run.world, "unit": "hom-mult/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "A_step_bytes_per_op": wl.a_limbs * 8 * wl.n, "pipeline_roofline": wl.pipeline_roofline(per_gpu, HBM_PEAK_GBS)}
    if launches:
        ent["roofline"] = roofline_entry(wl.family, wl.alg_bytes_per_step, steps, launches, kern_ms, dt, wl.logn, True, sclk_mhz=run.sclk())
    ent["step"] = step_kernels(run, wl, "B")
    if lib is not None:
        ok, cnt, classes = wl.verify(lib)
        ent["verified"] = bool(ok)
        ent["verified_outputs"] = cnt
        ent["checker_evaluations"] = classes
    return ent


def level_a_section(run: Run, lib, wl):
    """The opt-in parity level A (hp_ctx_set_parity_level; SURVEY.md section 8 "Parity levels"): the SAME C3 and C5 buffers and
    calls as the headline / the bgv section with the transforms of the pipelines on the FP64 residue kernels (hp_ntt_a.hip).
    Every output word must equal reduce_strict of the checker's raw word.  The headline `value` is never taken from here."""
    eng = run.eng
    out = {"what": "ckks::mult + relinearize + rescale_inplace (C3) and bgv mult + relinearize + mod_switch (C5 per GPU) with "
                   "hp_ctx_set_parity_level(HP_PARITY_A): canonical residues instead of hehub's lazy representatives; NTT / "
                   "mod-arith primitives unaffected"}
    eng.set_parity_level("A")
    try:
        for key, w, steps in (("ckks", wl, run.args.steps), ("bgv", Scheme(run, "bgv", 0, 0, 3, seed=5), max(run.args.steps, 20))):
            dt, launches, kern_ms = timed_launches(run, w.step, w.family, steps, warm=3)
            per_gpu = w.B * steps / dt
            ent = {"workload": w.cfg["workload"], "per_s": per_gpu * run.world, "unit": "hom-mult/s", "steps": steps,
                   "ms_per_step": 1e3 * dt / steps, "pipeline_roofline": w.pipeline_roofline(per_gpu, HBM_PEAK_GBS)}
            if launches:
                r = roofline_entry(w.family, w.alg_bytes_per_step, steps, launches, kern_ms, dt, w.logn, True, level="A", sclk_mhz=run.sclk())
                r["kernel"] = "k_ntt_fwd_a (register/LDS-tiled forward NTT, FP64 residue butterflies), digit-spread launch"
                ent["roofline"] = r
            ent["step"] = step_kernels(run, w, "A")
            if lib is not None:
                ok, cnt, classes = w.verify(lib, strict=True)
                ent.update({"verified": bool(ok), "verified_outputs": cnt, "checker_evaluations": classes,
                            "verify_what": "every output word == reduce_strict(checker's raw word)"})
            out[key] = ent
    finally:
        eng.set_parity_level("B")
    out["ntt"] = residue_transform_rates(run, lib)
    return out


def residue_transform_rates(run: Run, lib):
    """the limb transforms as residues (hp_dev_ntt_residues / hp_dev_intt_residues: canonical words through the FP64 kernels) at the
    shapes of the `ntt` and `c2` sections: N = 32768 over the C3 ciphertext moduli (one 50-bit, nine 40-bit) at the C3 batch's limb
    count and in steady state, and BASELINE config 2 exactly (four 50-bit moduli: every limb takes the extra range reductions)"""
    import numpy as np

    P, eng, steps = run.P, run.eng, run.args.steps
    out = {"what": "forward: every word == the reference's lazy word modulo q; inverse: the words of intt_negacyclic_inplace (ntt.h:88-92)"}
    for key, logn, moduli, B in (("32768", 15, P.C3_Q, 512), ("steady_32768", 15, P.C3_Q, 2560), ("c2", P.C2_LOGN, P.C2_MODULI, P.C2_BATCH)):
        n, L = 1 << logn, len(moduli)
        xb = Batch(run.torch, B, (L, n), moduli, run.dev, 140 + logn + 100 * run.rank, 3)
        x = xb.full
        ent = {"N": n, "limbs_per_launch": B * L, "moduli_bits": sorted({int(q).bit_length() for q in moduli})}
        for name, fam, fn in (("forward", "ntt", lambda: eng.ntt_residues_(moduli, x)), ("inverse", "intt", lambda: eng.intt_residues_(moduli, x))):
            dt, launches, kern_ms = timed_launches(run, fn, fam, steps, warm=2)
            ent[name] = rate_entry(B * L, 16.0 * n, steps, run.world, dt, launches, kern_ms)
            ent[name]["unit"] = "limb-NTT/s"
        if lib is not None:
            idx, host = xb.classes()
            qcol = np.array(moduli, dtype=np.uint64)[:, None]
            y = xb.fresh()
            eng.ntt_residues_(moduli, y)
            fwd = np.stack([lib.poly_ntt(moduli, host[c]) for c in range(len(idx))])
            ok1, cnt = compare_classes(run.torch, y, fwd % qcol, xb.period, idx)
            eng.intt_residues_(moduli, y)                                   # (input: the canonical forward words)
            inv = np.stack([lib.poly_reduce_strict(moduli, lib.poly_intt(moduli, fwd[c] % qcol)) for c in range(len(idx))])
            ok2, _ = compare_classes(run.torch, y, inv, xb.period, idx)
            ent["verified"], ent["verified_polynomials"] = bool(ok1 and ok2), cnt
        out[key] = ent
        del x, xb
    return out


def hbm_copy_ceiling(run: Run):
    """The measured stream ceiling beside the 8 TB/s spec peak: 1 GiB -> 1 GiB device-to-device, read + write
    bytes over the kernel time, (a) the engine's own copy kernel (hp_dev_copy: 16 bytes per lane, non-temporal -- the access
    pattern of every coefficient-wise kernel) from the library's HIP events, (b) the HIP runtime's device-to-device copy
    (torch copy_ = hipMemcpyDtoDAsync) between torch events on the same stream.  The ceiling is the better of the two."""
    torch, eng = run.torch, run.eng
    words = 1 << 27
    src = torch.empty(words, dtype=torch.int64, device=run.dev).random_()
    dst = torch.empty_like(src)
    steps = 20
    dt, launches, kern_ms = timed_launches(run, lambda: eng.copy(src, out=dst), "copy", steps, warm=3)
    ok = bool(torch.equal(src, dst))
    ours = 2.0 * 8 * words * launches / (kern_ms * 1e-3) / 1e9 if launches else None
    dst.zero_()
    for _ in range(3):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    rt = 2.0 * 8 * words * steps / (a.elapsed_time(b) * 1e-3) / 1e9
    best = max(v for v in (ours, rt) if v)
    # A 1 read : 1 write copy is not the fastest stream HBM serves: the read / write mixes of the engine's own streaming kernels at
    # the same footprint (2 reads : 1 write = a coefficient-wise addition, rns.cpp:58-87; 4 reads : 3 writes = the tensor product,
    # ckks/arith.cpp:55-62), library events around each launch, algorithmic bytes over the kernel time
    del src, dst
    n, L, rows = 1 << 15, 8, 512                      # 3 x 1 GiB for the addition, 7 x 0.5 GiB for the tensor product
    moduli = run.P.P40[:L]
    mixes = {"1R:1W": ours}
    a = rand_words(torch, (rows, L, n), moduli, run.dev, 91)
    b = rand_words(torch, (rows, L, n), moduli, run.dev, 92)
    o = torch.empty_like(a)
    dt, launches, kern_ms = timed_launches(run, lambda: eng.poly_add(moduli, a, b, out=o), "elem", steps, warm=3)
    mixes["2R:1W"] = 3.0 * 8 * rows * L * n * launches / (kern_ms * 1e-3) / 1e9 if launches else None
    del o
    B = rows // 4
    c1, c2 = a.view(-1)[:B * 2 * L * n].view(B, 2, L, n), b.view(-1)[:B * 2 * L * n].view(B, 2, L, n)
    dt, launches, kern_ms = timed_launches(run, lambda: eng.mult_low_level(moduli, c1, c2), "tensor", steps, warm=3)
    mixes["4R:3W"] = 7.0 * 8 * B * L * n * launches / (kern_ms * 1e-3) / 1e9 if launches else None
    stream_best = max(v for v in list(mixes.values()) + [rt] if v)
    return {"hbm_copy_ceiling_GBps": best, "frac_of_spec_peak": best / HBM_PEAK_GBS, "bytes_per_copy": 8 * words,
            "engine_copy_kernel_GBps": ours, "engine_copy_verified": ok, "hip_memcpy_d2d_GBps": rt,
            "counts": "read + write bytes", "steps": steps,
            "stream_mix_GBps": mixes, "hbm_stream_ceiling_GBps": stream_best,
            "note": "hbm_copy_ceiling_GBps is the 1 read : 1 write rate only; kernels with more reads than writes stream faster -- "
                    "compare a streaming kernel with the mix of its own shape (stream_mix_GBps), the best of which is hbm_stream_ceiling_GBps"}


def object_api_section(run: Run):
    """Throughput through hehub's OBJECT interface (ckks.h:270-313 is one ciphertext per call): examples/independent_mults, a C++
    program against hehub_amd/host/hehub.hpp, run as a child process AFTER the timed region on the same GPU (this process idles):
    B = 256 independent C3 hom-mults as the loop of single calls (over the layer's default lanes), as ONE batched call
    (hehub_amd_ext.hpp: amd::mult_rescale), and 8 independent chains of single calls on 1 lane / 8 lanes.  Every mode prints an
    FNV digest of all result words; `digests_equal` says the batched call and the lanes returned the words of the single calls
    (tests/test_object_api.py holds those to hehub on the CPU).  Not part of `value`."""
    import os
    import re
    import subprocess
    import time

    from hehub_amd.build import build_example

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P = run.P
    shape = [P.C3_LOGN, len(P.C3_Q), 256]
    t0 = time.perf_counter()
    # (1) the calls run one by one as they are made: HEHUB_AMD_DEFER=0, the escape from the layer's default
    out = subprocess.run([build_example("independent_mults")] + [str(a) for a in shape + ["all", 3, 8, 8, 6]], capture_output=True,
                         text=True, timeout=600, cwd=root, env=dict(os.environ, HEHUB_AMD_DEFER="0"))
    ent = {"program": "examples/independent_mults " + " ".join(str(a) for a in shape) + " all 3 8 8 6", "wall_s": round(time.perf_counter() - t0, 1),
           "N": 1 << shape[0], "L": shape[1], "B": shape[2], "unit": "hom-mult/s"}
    if out.returncode != 0:
        ent["error"] = (out.stdout[-300:] + out.stderr[-300:])
        ent["verified"] = False
        return ent
    dg = {}
    for line in out.stdout.splitlines():
        m = re.match(r"([\w-]+) digest (\w+)", line)
        if m:
            dg[m.group(1)] = m.group(2)
        m = re.match(r"serial ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\); the calls themselves returned after ([\d.]+) ms", line)
        if m:
            ent["single_calls"] = {"per_s": float(m.group(2)), "ms_per_hom_mult": float(m.group(1)), "host_ms_per_hom_mult": float(m.group(3)),
                                   "what": "HEHUB_AMD_DEFER=0: for i: ckks::mult(a[i], b[i], key); ckks::rescale_inplace(.) -- every call runs when it is made, default lanes"}
        m = re.match(r"batch ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\); first call .* ([\d.]+) ms per hom-mult", line)
        if m:
            ent["batched_call"] = {"per_s": float(m.group(2)), "ms_per_hom_mult": float(m.group(1)), "first_call_ms_per_hom_mult": float(m.group(3)),
                                   "what": "amd::mult_rescale(std::vector<CkksCt>, std::vector<CkksCt>, key): one engine call"}
        m = re.match(r"batch-chain rotate\+add\+rescale ([\d.]+) ms per ciphertext", line)
        if m:
            ent["batched_chain_ms_per_ct"] = float(m.group(1))
        m = re.match(r"chains (\d+) x (\d+) mult\+rotate: ([\d.]+) ms per step on 1 lane, ([\d.]+) ms on (\d+) lanes", line)
        if m:
            ent["independent_chains"] = {"chains": int(m.group(1)), "steps": int(m.group(2)), "ms_per_step_1_lane": float(m.group(3)),
                                         "ms_per_step_lanes": float(m.group(4)), "lanes": int(m.group(5)),
                                         "speedup": float(m.group(3)) / float(m.group(4)),
                                         "what": "x[c] = ckks::rotate(ckks::mult(x[c], b[c], key), key, 1), interleaved call by call"}
    ent["digests"] = dg
    ent["digests_equal"] = bool(dg) and dg.get("serial") == dg.get("batch") and dg.get("serial-chain") == dg.get("batch-chain") and \
        dg.get("chains") == dg.get("chains-lanes")
    # (2) the same program, unchanged, with NOTHING in the environment: the layer's default (calls are recorded and run as batches, hehub_amd/host/deferred_*.cpp)
    t0 = time.perf_counter()
    env0 = {k: v for k, v in os.environ.items() if k != "HEHUB_AMD_DEFER"}
    out2 = subprocess.run([build_example("independent_mults")] + [str(a) for a in shape + ["all", 3, 1, 8, 6]], capture_output=True,
                          text=True, timeout=600, cwd=root, env=env0)
    de = {"wall_s": round(time.perf_counter() - t0, 1), "what": "no environment variable (the layer's default): the scheme-level calls are recorded and run "
          "grouped as batched engine calls when somebody needs words; mult + rescale_inplace triples as the fused one-call pipeline"}
    dg2 = {}
    if out2.returncode == 0:
        for line in out2.stdout.splitlines():
            m = re.match(r"([\w-]+) digest (\w+)", line)
            if m:
                dg2[m.group(1)] = m.group(2)
            m = re.match(r"serial ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\)", line)
            if m:
                de["single_calls"] = {"per_s": float(m.group(2)), "ms_per_hom_mult": float(m.group(1))}
            m = re.match(r"chains (\d+) x (\d+) mult\+rotate: ([\d.]+) ms per step on 1 lane", line)
            if m:
                de["independent_chains_ms_per_step"] = float(m.group(3))
                if "independent_chains" in ent:
                    de["independent_chains_speedup_vs_eager_1_lane"] = ent["independent_chains"]["ms_per_step_1_lane"] / float(m.group(3))
            m = re.match(r"deferred: (\d+) recorded calls ran as (\d+) batched engine calls \((\d+) ", line)
            if m:
                de.update({"recorded_calls": int(m.group(1)), "batched_engine_calls": int(m.group(2)), "fused_triples": int(m.group(3))})
        de["digests_equal_eager"] = all(dg2.get(k) == dg.get(k) for k in ("serial", "serial-chain", "batch", "batch-chain", "chains"))
    else:
        de["error"] = (out2.stdout[-300:] + out2.stderr[-300:])
        de["digests_equal_eager"] = False
    ent["deferred"] = de
    # the headline of this section: hehub's loop of single calls, source and environment unchanged (ckks.h:270-313)
    ent["unchanged_loop"] = {"per_s": de.get("single_calls", {}).get("per_s"), "fused_triples": de.get("fused_triples", 0), "environment": "none",
                             "digest_equal": bool(dg2) and dg2.get("serial") == dg.get("serial"),
                             "what": "for i: ckks::mult(a[i], b[i], key); ckks::rescale_inplace(.), B = 256, as hehub's callers write it"}
    ent["verified"] = ent["digests_equal"] and de["digests_equal_eager"]
    # (3) the same program over DEVICE RANKS (HEHUB_AMD_DEVICES: the layer places every call on a device itself): every GPU this process
    # can see, or -- on a one-GPU box -- two ranks sharing GPU 0 (the code path, not a scaling number; the entry says which)
    try:
        ent["devices"] = devices_entry(run, root, shape, dg)
        ent["verified"] = ent["verified"] and ent["devices"]["verified"]
    except Exception as e:   # noqa: BLE001
        ent["devices"] = {"error": repr(e)[:300], "verified": None}
    try:
        ent["matvec"] = matvec_entry(root)
        ent["verified"] = ent["verified"] and ent["matvec"]["verified"]
    except Exception as e:   # noqa: BLE001
        ent["matvec"] = {"error": repr(e)[:300], "verified": None}
    try:
        ent["reference_benchmark"] = reference_benchmark_entry(root)
        ent["verified"] = ent["verified"] and ent["reference_benchmark"]["verified"]
    except Exception as e:   # noqa: BLE001
        ent["reference_benchmark"] = {"error": repr(e)[:300], "verified": None}
    return ent


def devices_entry(run: Run, root: str, shape, want):
    """hehub's object API over device ranks: examples/independent_mults with HEHUB_AMD_DEVICES and nothing else in the environment --
    the unchanged loop of single calls (recorded, grouped per rank), the batched form (contiguous slices per rank), chains; digests must be
    the one-device digests `want`; the layer's own counters say how many engine calls each rank made and how many operands had to move"""
    import os
    import re
    import subprocess
    import time

    from hehub_amd.build import build_example

    visible = run.torch.cuda.device_count()
    shared = visible < 2
    ranks = 2 if shared else min(visible, 8)
    devs = ",".join(["0"] * ranks) if shared else str(ranks)
    B = shape[2] if shared else shape[2] * ranks      # (shared: the one GPU's batch; real devices: the per-GPU batch on each)
    args = [str(a) for a in [shape[0], shape[1], B, "all", 3, 1, 8 * ranks, 6]]
    env = {k: v for k, v in os.environ.items() if k not in ("HEHUB_AMD_DEFER", "HEHUB_AMD_LANES")}
    t0 = time.perf_counter()
    out = subprocess.run([build_example("independent_mults")] + args, capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, HEHUB_AMD_DEVICES=devs))
    ent = {"program": "HEHUB_AMD_DEVICES=" + devs + " examples/independent_mults " + " ".join(args), "ranks": ranks, "ranks_share_one_gpu": shared,
           "B": B, "wall_s": round(time.perf_counter() - t0, 1), "unit": "hom-mult/s"}
    if out.returncode != 0:
        ent.update(error=(out.stdout[-300:] + out.stderr[-300:]), verified=False)
        return ent
    dg = {m.group(1): m.group(2) for m in re.finditer(r"^([\w-]+) digest (\w+)", out.stdout, re.M)}
    m = re.search(r"^serial ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\)", out.stdout, re.M)
    if m:
        ent["unchanged_loop_per_s"] = float(m.group(2))
    m = re.search(r"^batch ([\d.]+) ms per hom-mult \((\d+) hom-mult/s\)", out.stdout, re.M)
    if m:
        ent["batched_call_per_s"] = float(m.group(2))
    m = re.search(r"devices (\d+) engine calls per device rank:((?: \d+)+); copies between ranks (\d+) \(([\d.]+) MiB\)", out.stdout)
    if m:
        ent["engine_calls_by_rank"] = [int(x) for x in m.group(2).split()]
        ent["copies_between_ranks"] = int(m.group(3))
        ent["MiB_between_ranks"] = float(m.group(4))
    # (the digests depend on the batch: only a run at the one-device batch can be compared with the one-device digests)
    ent["digests_equal_one_device"] = (all(dg.get(k) == want.get(k) for k in ("serial", "batch", "serial-chain", "batch-chain")) if B == shape[2] else None)
    ent["modes_agree"] = bool(dg) and dg.get("serial") == dg.get("batch") and dg.get("serial-chain") == dg.get("batch-chain")
    ent["verified"] = ent["modes_agree"] and ent["digests_equal_one_device"] is not False and \
        all(c > 0 for c in ent.get("engine_calls_by_rank", [0]))
    ent["summary"] = {"ranks": ranks, "shared_gpu": shared, "unchanged_loop_per_s": ent.get("unchanged_loop_per_s"), "batched_per_s": ent.get("batched_call_per_s")}
    return ent


def reference_benchmark_entry(root: str):
    """hehub's OWN benchmark (bench/benchmarks.cpp:21-37): ckks::rotate(ct, rot_key, 1), one ciphertext per call, at its four parameter
    sets (N = 2^12 .. 2^15 with the modulus chains of ckks::create_params(N, scaling_bits): 2 x 36 .. 15 x 55 bits) -- the loop on synthetic
    words through hehub's object API (examples/rotate_bench as a child process, default lanes): ms per rotation with a look at every result
    before the next call (hehub's synchronous semantics) and back to back, recorded (HEHUB_AMD_DEFER=1) as a second run; every digest must be
    hehub's own (tests/golden/rotate_bench.json, generated from hehub on the CPU).  `cpu_reference_ms`: the prebuilt
    oracle/_ref/ref_rotbench_cpu (hehub itself, one core) on this host, when it travelled."""
    import json
    import os
    import subprocess
    import sys
    import time

    from hehub_amd.build import build_example

    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    from make_rotate_bench import LOGNS, REF, run

    with open(os.path.join(root, "tests", "golden", "rotate_bench.json")) as f:
        want = {int(k): v for k, v in json.load(f)["digests"].items()}
    t0 = time.perf_counter()
    binary = build_example("rotate_bench")
    ent = {"program": "examples/rotate_bench 100", "unit": "ms per rotation", "what": "bench/benchmarks.cpp:21-37 on synthetic words, one ciphertext per call",
           "by_N": {}}
    ok = True
    eager, _ = run(binary, 100, 0, {"HEHUB_AMD_DEFER": "0"})
    deferred, _ = run(binary, 100, 0, {"HEHUB_AMD_DEFER": "1"})
    cpu = {}
    if os.path.exists(REF):
        cpu, _ = run(REF, 2)
    for lg in LOGNS:
        row = {"hehub_digest": want[lg]}
        if lg in eager:
            row.update({"look_after_every_call": eager[lg][1], "back_to_back": eager[lg][2]})
        if lg in deferred:
            row["recorded_back_to_back"] = deferred[lg][2]
        row["digests_equal"] = lg in eager and lg in deferred and eager[lg][0] == want[lg] and deferred[lg][0] == want[lg]
        ok = ok and row["digests_equal"]
        if lg in cpu:
            row["cpu_reference_ms"] = cpu[lg][1]
            row["cpu_reference_digest_equal"] = cpu[lg][0] == want[lg]
            ok = ok and row["cpu_reference_digest_equal"]
            if lg in eager:
                row["speedup_vs_cpu_reference"] = cpu[lg][1] / eager[lg][1]
        ent["by_N"][str(1 << lg)] = row
    ent["wall_s"] = round(time.perf_counter() - t0, 1)
    ent["verified"] = ok
    return ent


def matvec_entry(root: str):
    """hehub's circuit-level caller of the key switch, the diagonal loop of matrix_vector_mul_short (src/circuits/linear_algebra.h:104-136),
    at the C3 shape, width 16 (30 rotations of one vector under 30 keys + 16 plaintext products): examples/diag_matvec as a child process --
    eager single calls over the lanes, deferred mode (the rotations run as ONE hp_dev_ckks_rotate_many_rows sequence with a key per
    ciphertext), the batched form amd::rotate(ct, keys, steps); every digest must be hehub's own (tests/golden/matvec.json, generated from
    hehub on the CPU).  `cpu_reference_ms`: the prebuilt oracle/_ref/ref_matvec_cpu (hehub itself) on this host, when it travelled."""
    import json
    import os
    import re
    import subprocess
    import time

    from hehub_amd.build import build_example

    case = [15, 10, 16, "short"]
    with open(os.path.join(root, "tests", "golden", "matvec.json")) as f:
        want = json.load(f)["digests"][" ".join(str(a) for a in case)]
    t0 = time.perf_counter()
    out = subprocess.run([build_example("diag_matvec")] + [str(a) for a in case + [5]], capture_output=True, text=True, timeout=600, cwd=root)
    ent = {"program": "examples/diag_matvec 15 10 16 short 5", "wall_s": round(time.perf_counter() - t0, 1), "N": 32768, "L": 10, "width": 16,
           "rotations": 30, "keys": 30, "unit": "ms per product vector", "hehub_digest": want}
    if out.returncode != 0:
        ent["error"] = (out.stdout[-300:] + out.stderr[-300:])
        ent["verified"] = False
        return ent
    dg = {m.group(1): m.group(2) for m in re.finditer(r"([\w-]+) digest (\w+)", out.stdout)}
    ms = {m.group(1): float(m.group(2)) for m in re.finditer(r"([\w-]+) ([\d.]+) ms per product vector", out.stdout)}
    ent["ms"] = ms
    ent["digests"] = dg
    ent["verified"] = len(dg) == 3 and all(v == want for v in dg.values())
    ref = os.path.join(root, "oracle", "_ref", "ref_matvec_cpu")
    if os.path.exists(ref):
        r = subprocess.run([ref] + [str(a) for a in case + [1]], capture_output=True, text=True, timeout=600, cwd=root)
        m = re.search(r"loop ([\d.]+) ms per product vector", r.stdout)
        d = re.search(r"loop digest (\w+)", r.stdout)
        if r.returncode == 0 and m and d:
            ent["cpu_reference_ms"] = float(m.group(1))
            ent["cpu_reference_digest_equal"] = d.group(1) == want
            if "deferred" in ms:
                ent["speedup_vs_cpu_reference"] = float(m.group(1)) / ms["deferred"]
    return ent
