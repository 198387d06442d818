"""The workloads of bench.py's timed region.  A workload owns its device buffers and says what one step is, which kernel
family dominates it, how many algorithmic bytes that family moves per step, and how to check the buffers the timed steps
wrote (verify() gets the CPU checker handed in by bench.py; nothing here imports it)."""
from __future__ import annotations

import os

from .inputs import Batch, compare_classes, rand_words
from .timing import Run

NAMES = ["ckks", "ntt", "ntt15", "intt", "intt15", "bgv", "rotate", "ckks-limb", "encdec", "mul", "add", "ckks-hks"]


class Workload:
    scaling = "weak"
    family = "ntt"          # the profiling family whose launches the roofline entry is about
    spread = False          # that family's launch is the digit-spread one (own PMC entry in profiles/traffic.json)
    a_limbs = None          # pipelines: compulsory bytes per op in limbs (S = 8N), SURVEY.md 8d
    verifiable = True
    cpu_name = None         # which CPU baseline sample stands beside it (bench.py cpu_baseline)

    def __init__(self, run: Run):
        self.run = run

    def step(self):
        raise NotImplementedError

    def verify(self, lib):
        """-> (ok, outputs compared, checker evaluations)"""
        raise NotImplementedError


class Transform(Workload):
    """BASELINE config 2 (ntt / intt: N = 16384, four 50-bit primes, 1024 polynomials) or the transform shape inside the C3
    pipeline (ntt15 / intt15: N = 32768, 10 moduli + special prime), in place"""

    def __init__(self, run, name, batch=0, logn=0, period=3, seed=2):
        super().__init__(run)
        P, torch, eng = run.P, run.torch, run.eng
        self.inverse = name.startswith("intt")
        if name.endswith("15"):
            lg, self.moduli, B = P.C3_LOGN, P.C3_MODULI_EXT, batch or 256
        else:
            lg, self.moduli, B = P.C2_LOGN, P.C2_MODULI, batch or P.C2_BATCH
        self.logn = logn or lg
        n, L = 1 << self.logn, len(self.moduli)
        self.n, self.L, self.B = n, L, B
        self.xb = Batch(torch, B, (L, n), self.moduli, run.dev, seed + run.rank, period)
        self.x = self.xb.full
        self.units_per_step = B * L
        self.family = "intt" if self.inverse else "ntt"
        self.alg_bytes_per_step = 16.0 * n * B * L
        self.metric, self.unit = "limb_ntt_per_s", "limb-NTT/s"
        shape = "C2" if (self.logn == 14 and L == 4) else "C3-shape" if self.logn == 15 else "custom"
        self.cfg = {"workload": f"{shape}: batched {'inverse' if self.inverse else 'forward'} negacyclic NTT, N={n}, {L} RNS limbs, "
                                f"batch={B} polynomials per GPU", "N": n, "limbs": L, "batch_per_gpu": B}
        self.cpu_name = name
        self.seed = seed

    def step(self):
        (self.run.eng.intt_ if self.inverse else self.run.eng.ntt_)(self.moduli, self.x)

    def verify(self, lib):
        # the timed buffer has been transformed in place steps+warmup times; check one fresh application of the same call
        import numpy as np

        run, xb = self.run, self.xb
        idx, host = xb.classes((0, self.B // 2, self.B - 1))
        y = xb.fresh() if xb.period else rand_words(run.torch, (self.B, self.L, self.n), self.moduli, run.dev, self.seed + run.rank)
        (run.eng.intt_ if self.inverse else run.eng.ntt_)(self.moduli, y)
        exp = np.stack([(lib.poly_intt if self.inverse else lib.poly_ntt)(self.moduli, host[c]) for c in range(len(idx))])
        ok, cnt = compare_classes(run.torch, y, exp, xb.period, idx)
        return ok, cnt, len(idx)


class Coeffwise(Workload):
    """coefficient-wise kernels at the C2 shape: RnsPolynomial operator* (hybrid Montgomery + Harvey product,
    rns.cpp:120-140) / operator+= (rns.cpp:58-87); 24*N algorithmic bytes per limb"""
    family = "elem"

    def __init__(self, run, name, batch=0, period=3):
        super().__init__(run)
        P, torch, eng = run.P, run.torch, run.eng
        self.op, self.logn, self.moduli = name, P.C2_LOGN, P.C2_MODULI
        B = batch or P.C2_BATCH
        n, L = 1 << self.logn, len(self.moduli)
        self.n, self.L, self.B = n, L, B
        self.ab = Batch(torch, B, (L, n), self.moduli, run.dev, 21 + run.rank, period)
        self.bb = Batch(torch, B, (L, n), self.moduli, run.dev, 22 + run.rank, period)
        self.out = eng.empty((B, L, n))
        self.units_per_step = B * L
        self.alg_bytes_per_step = 24.0 * n * B * L
        self.metric, self.unit = f"limb_{name}_per_s", "limb-op/s"
        self.cfg = {"workload": f"C2 shape: coefficient-wise modular {'multiply' if name == 'mul' else 'add'}, N={n}, {L} RNS limbs, "
                                f"batch={B} polynomials per GPU", "N": n, "limbs": L, "batch_per_gpu": B}
        self.cpu_name = name

    def step(self):
        f = self.run.eng.poly_mul if self.op == "mul" else self.run.eng.poly_add
        f(self.moduli, self.ab.full, self.bb.full, out=self.out)

    def verify(self, lib):
        import numpy as np

        sample = (0, self.B // 2, self.B - 1)
        idx, ha = self.ab.classes(sample)
        _, hb = self.bb.classes(sample)
        f = lib.poly_mul if self.op == "mul" else lib.poly_add
        exp = np.stack([f(self.moduli, ha[c], hb[c]) for c in range(len(idx))])
        ok, cnt = compare_classes(self.run.torch, self.out, exp, self.ab.period, idx)
        return ok, cnt, len(idx)


class HksMult(Workload):
    """EXTENSION, not comparable with the reference: the C3 ciphertext chain with a hybrid key switch (digits of alpha
    moduli, k special primes); keys in the hybrid format, results differ from hehub's by design"""
    verifiable = False
    cpu_name = "ckks"

    def __init__(self, run, batch=0, logn=0, alpha=2, k=2):
        super().__init__(run)
        P, torch, eng = run.P, run.torch, run.eng
        self.logn = logn or P.C3_LOGN
        L = len(P.C3_Q)
        self.mext = P.C3_Q + P.ntt_primes(k, P.C3_LOGN, 50, exclude=P.C3_Q)
        B = batch or P.C3_BATCH
        n, nd = 1 << self.logn, (L + alpha - 1) // alpha
        self.n, self.L, self.B, self.k, self.alpha = n, L, B, k, alpha
        self.ct1 = rand_words(torch, (B, 2, L, n), self.mext[:L], run.dev, 3 + run.rank)
        self.ct2 = rand_words(torch, (B, 2, L, n), self.mext[:L], run.dev, 1003 + run.rank)
        self.key = rand_words(torch, (nd, 2, L + k, n), self.mext, run.dev, 7)
        self.out = eng.empty((B, 2, L - 1, n))
        self.units_per_step = B
        fwd = nd * (L + k) - L                            # lifted-digit transforms per ciphertext: the one k_ntt_fwd launch per step
        self.alg_bytes_per_step = 16.0 * n * fwd * B      # (ModDown and rescale transforms are k_ntt_fwd_drop launches, family "ntt_drop")
        self.metric, self.unit = "ckks_hks_hom_mult_per_s", "hom-mult/s"
        self.cfg = {"workload": f"EXTENSION (not hehub-compatible keys): ckks mult + hybrid-key relinearisation (digits of {alpha} moduli, "
                                f"{k} special primes) + rescale, N={n}, L={L}, batch={B} ciphertext pairs per GPU",
                    "N": n, "L": L, "batch_per_gpu": B, "hks_alpha": alpha, "hks_k": k, "digits": nd,
                    "forward_transforms_per_op": fwd + 2 * L + 2 * (L - 1), "reference_algorithm_forward_transforms_per_op": L * L + 4 * L - 2}

    def step(self):
        self.run.eng.ckks_mult_hks(self.mext, self.k, self.alpha, self.ct1, self.ct2, self.key, out=self.out)


class EncDec(Workload):
    """either side of the path: encrypt_core on caller-supplied samples, then decrypt_core"""
    verifiable = False
    cpu_name = "encdec"

    def __init__(self, run, batch=0):
        super().__init__(run)
        P, torch = run.P, run.torch
        self.logn, self.moduli = P.C3_LOGN, P.C3_Q
        B = batch or P.C3_BATCH
        n, L = 1 << self.logn, len(self.moduli)
        self.n, self.L, self.B = n, L, B
        self.c1 = rand_words(torch, (B, L, n), self.moduli, run.dev, 11 + run.rank)
        self.pt = rand_words(torch, (B, L, n), self.moduli, run.dev, 12 + run.rank)
        self.sk = rand_words(torch, (L, n), self.moduli, run.dev, 13)
        self.noise = torch.randint(-19, 20, (B, n), dtype=torch.int64, device=run.dev)
        self.units_per_step = B
        self.alg_bytes_per_step = 16.0 * n * 2 * L * B      # the two forward transforms per ciphertext (noise, plaintext)
        self.metric, self.unit = "rlwe_encrypt_decrypt_per_s", "ciphertext/s"
        self.cfg = {"workload": f"C3 shape: rlwe encrypt_core (given samples) + decrypt_core, N={n}, L={L}, batch={B} ciphertexts per GPU",
                    "N": n, "L": L, "batch_per_gpu": B, "A_step_bytes_per_op": (15 * L + 1) * 8 * n}

    def step(self):
        eng = self.run.eng
        eng.rlwe_decrypt_core(self.moduli, eng.rlwe_encrypt_core(self.moduli, self.noise, self.c1, self.pt, self.sk), self.sk)


class Scheme(Workload):
    """ckks (BASELINE config 3/4: mult + relinearize + rescale, N = 32768, L = 10), bgv (config 5 per-GPU shape: mult +
    relinearize + mod_switch, N = 8192, L = 6, t = 65537), rotate (the reference's own benchmark, bench/benchmarks.cpp:21-37,
    at the C3 shape) and ckks-limb (one small batch cut by output modulus over all ranks)"""
    spread = True

    def __init__(self, run, name, batch=0, logn=0, period=3, seed=3):
        super().__init__(run)
        P, torch, eng = run.P, run.torch, run.eng
        self.name = name
        if name in ("ckks", "rotate", "ckks-limb"):
            lg, self.mext, self.t, B0 = P.C3_LOGN, P.C3_MODULI_EXT, 0, (P.C3_BATCH if name != "ckks-limb" else 8)
        else:
            lg, self.mext, self.t, B0 = P.C5_LOGN, P.C5_MODULI_EXT, P.C5_T, P.C5_BATCH // 8
        B = batch or B0
        self.logn = (logn or lg) if name in ("ckks", "rotate") else lg
        n, L = 1 << self.logn, len(self.mext) - 1
        self.n, self.L, self.B = n, L, B
        shared = name == "ckks-limb"            # the limb-sharded mode works on the SAME batch on every rank
        self.b1 = Batch(torch, B, (2, L, n), self.mext[:L], run.dev, seed + (0 if shared else run.rank), period)
        self.b2 = Batch(torch, B, (2, L, n), self.mext[:L], run.dev, seed + 1000 + (0 if shared else run.rank), period)
        self.key = rand_words(torch, (L, 2, L + 1, n), self.mext, run.dev, 7)
        self.out = eng.empty((B, 2, L - 1, n))
        self.result = self.out
        self.units_per_step = B
        # compulsory bytes per op in limbs (S = 8N): SURVEY.md 8d for hom-mult; for a rotation the tensor product
        # (7L) becomes the gather (4L), there is no second drop and only c0 gets the moved addend
        self.a_limbs = 5 * L * L + 36 * L
        self.metric, self.unit = "ckks_hom_mult_per_s", "hom-mult/s"
        self.cpu_name = "ckks"
        if name == "ckks-limb":
            # latency mode (hehub_amd/sharded.py): the SAME small batch on every rank, cut by output modulus, with
            # the exchange of the key-switch digits between the ranks; total work is fixed as N grows -> strong scaling
            from hehub_amd.sharded import Comm, ShardedMult

            self.comm, self.sm = Comm(transport=getattr(run.args, "limb_transport", None)), ShardedMult(eng, self.mext, run.world)
            self.bufs = self.sm.buffers(B, n)
            title = "C3 shape, limb-sharded latency mode: ckks::mult + relinearize + rescale_inplace"
            self.scaling = "strong"
            self.units_per_step = B / run.world   # `value` multiplies by world: the batch is shared, not replicated work
            self.transport = self.comm.transport
        elif name == "rotate":
            self.metric, self.unit = "ckks_rotation_per_s", "rotation/s"
            title = "C3 shape: ckks::rotate (gather + key switch + drop of the special prime)"
            self.a_limbs = 5 * L * L + 22 * L + 6
            self.cpu_name = "rotate"
        elif name == "ckks":
            title = "C3: ckks::mult + relinearize + rescale_inplace"
        else:
            self.metric = "bgv_hom_mult_per_s"
            title = "C5 shape: bgv mult_low_level + relinearize + mod_switch_inplace"
            self.cpu_name = "bgv"
        # the dominant kernel is k_ntt_fwd in its digit-spread launch (rgsw.cpp:108-119): L*L limb transforms per
        # ciphertext, one launch per step; the fused drop-last-prime launches are a different kernel (k_ntt_fwd_drop,
        # profiling family "ntt_drop") and are not mixed into this roofline
        self.alg_bytes_per_step = 16.0 * n * L * L * B
        self.cfg = {"workload": f"{title}, N={n}, L={L} moduli + special prime, batch={B} ciphertext pairs per GPU",
                    "N": n, "L": L, "batch_per_gpu": B, "sub_batch": int(os.environ.get("HP_MULT_CHUNK", "0")) or B,
                    "input_period": self.b1.period, "A_step_bytes_per_op": self.a_limbs * 8 * n}
        if self.t:
            self.cfg["plain_modulus"] = self.t
        if name == "ckks-limb":
            self.cfg["digit_exchange"] = self.transport

    def step(self):
        eng, ct1, ct2 = self.run.eng, self.b1.full, self.b2.full
        if self.name == "ckks-limb":
            self.result = self.sm.run(self.comm, ct1, ct2, self.key, self.bufs)
        elif self.name == "rotate":
            self.result = eng.ckks_rotate(self.mext, ct1, self.key, 1)
        elif self.name == "ckks":
            eng.ckks_mult(self.mext, ct1, ct2, self.key, out=self.out)
        else:
            eng.bgv_mult(self.mext, self.t, ct1, ct2, self.key, out=self.out)

    def _check(self, lib, c1, c2, k):
        if self.name == "rotate":
            return lib.ckks_rotate(self.mext, c1, k, 1)
        if self.name == "bgv":
            return lib.bgv_mult(self.mext, self.t, c1, c2, k)
        return lib.ckks_mult(self.mext, c1, c2, k)

    def verify(self, lib, strict=False):
        # the buffer the LAST timed step wrote, every ciphertext of it; strict: the engine ran at parity level A, whose outputs are
        # reduce_strict (mod_arith.h:58-72) of the checker's raw words
        import numpy as np

        B = self.B
        sample = (0, 1, B // 2, B - 1)
        idx, h1 = self.b1.classes(sample)
        _, h2 = self.b2.classes(sample)
        hk = self.key.cpu().numpy().view(np.uint64)
        exp = np.stack([self._check(lib, h1[c], h2[c], hk) for c in range(len(idx))])
        if strict:
            q = np.array(self.mext[:exp.shape[-2]], dtype=np.uint64)[:, None]
            exp = np.where(exp >= q, exp - q, exp)
        res = self.result
        if res.shape[0] != B:
            return False, 0, len(idx)
        ok, cnt = compare_classes(self.run.torch, res, exp, self.b1.period, idx)
        return ok, cnt, len(idx)

    def family_limbs(self):
        """A_step split by profiling family (limbs of S = 8N bytes per op): tensor 7L; inverse transforms 2L (the key
        switch's coefficient rows) + 2 x 2 per drop; digit transforms 2L^2; inner product (L + 1)(3L + 2); the two fused drops
        12L + 10(L - 1).  A rotation has the gather instead of the tensor product and one drop.  Sums to a_limbs."""
        L = self.L
        if self.name == "rotate":
            return {"elem": 4 * L, "intt": 2 * L + 4, "ntt": 2 * L * L, "ks_inner": (L + 1) * (3 * L + 2), "ntt_drop": 12 * L - 2 * L}
        return {"tensor": 7 * L, "intt": 2 * L + 8, "ntt": 2 * L * L, "ks_inner": (L + 1) * (3 * L + 2), "ntt_drop": 12 * L + 10 * (L - 1)}

    def pipeline_roofline(self, per_gpu_ops_per_s, peak_gbs):
        """A_step / A_prim / A_min fractions of a per-GPU op rate"""
        n, L, B = self.n, self.L, self.B
        a_step = self.a_limbs * 8 * n
        out = {"A_step_GBps": per_gpu_ops_per_s * a_step / 1e9, "frac_of_hbm_peak": per_gpu_ops_per_s * a_step / 1e9 / peak_gbs}
        if self.name in ("ckks", "bgv"):   # every primitive its own pass / I-O lower bound
            a_prim = (6 * L * L + 67 * L) * 8 * n
            a_min = (6 * L - 2) * 8 * n + 2 * L * (L + 1) * 8 * n / B
            out.update({"A_prim_frac_of_hbm_peak": per_gpu_ops_per_s * a_prim / 1e9 / peak_gbs,
                        "A_min_frac_of_hbm_peak": per_gpu_ops_per_s * a_min / 1e9 / peak_gbs})
        return out


def make(run: Run, name: str):
    a = run.args
    if name in ("ntt", "ntt15", "intt", "intt15"):
        return Transform(run, name, a.batch, a.logn, a.input_period)
    if name in ("mul", "add"):
        return Coeffwise(run, name, a.batch, a.input_period)
    if name == "ckks-hks":
        return HksMult(run, a.batch, a.logn, a.hks_alpha, a.hks_k)
    if name == "encdec":
        return EncDec(run, a.batch)
    return Scheme(run, name, a.batch, a.logn, a.input_period)
