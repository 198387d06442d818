"""Case list shared by the golden-vector generator and the golden test.

`run_cases(lib)` evaluates every case with one implementation (the compiled
reference when generating, the C oracle in the CPU tier, the HIP engine behind
tests/test_gpu_golden.py's adapter in the GPU tier) on inputs derived from
fixed splitmix64 seeds, and returns {case name: summary}.  A summary holds the
FNV-1a-64 digest of the raw little-endian output words, the first and last 8
words, and the full output when it is small.  `fnv` is the digest function
(bytes of a contiguous array -> int); the default is the oracle library's, the
engine test passes the product's own (hp_wire_fnv1a64) so that no checker
library is loaded there.  Cases an implementation has no entry point for
(the scalar known answers) are skipped when the method is absent.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import params as P  # noqa: E402
from splitmix import SplitMix  # noqa: E402

U = np.uint64
_fnv = None


def _default_fnv(flat):
    global _fnv
    if _fnv is None:
        from oracle.pyoracle import Oracle

        _fnv = Oracle("orc").fnv  # digest helper only (fnv1a64 is not reference code)
    return _fnv(flat)


def summary(a: np.ndarray) -> dict:
    flat = np.ascontiguousarray(a).reshape(-1)
    s = {"n": int(flat.size), "fnv": f"{(_digest or _default_fnv)(flat):016x}",
         "head": [int(x) for x in flat[:8]], "tail": [int(x) for x in flat[-8:]]}
    if flat.size <= 256:
        s["full"] = [int(x) for x in flat]
    return s


_digest = None


def run_cases(lib, big: bool = True, fnv=None) -> dict:
    global _digest
    _digest = fnv
    try:
        return _run_cases(lib, big)
    finally:
        _digest = None


def _run_cases(lib, big: bool) -> dict:
    out = {}
    # ---- scalar known answers -------------------------------------------------
    q40 = P.P40[0]
    if hasattr(lib, "mul_mod_harvey_lazy"):
        out["harvey_scalar"] = {"value": lib.mul_mod_harvey_lazy(q40, 123456789012345678, 987654321,
                                                                 (987654321 << 64) // q40)}
        out["inverse_65537_mod_q40"] = {"value": lib.inverse_mod_prime(65537, q40)}
        for q, logn in ((P.C1_Q, 12), (q40, 14), (65537, 4), (P.P50[0], 15)):
            out[f"psi_q{q}_logn{logn}"] = {"value": lib.unity_root(q, 1 << logn)}

    # ---- batched modular arithmetic (tests/mod_arith_t.cpp moduli + list primes)
    for q in P.BARRETT_TEST_Q + [P.P40[0], P.P50[0], P.C1_Q]:
        a = SplitMix(11).words(2048, 0)
        b = SplitMix(12).words(2048, 0)
        out[f"barrett_lazy_q{q}"] = summary(lib.batched_barrett_lazy(q, a))
        out[f"barrett_strict_q{q}"] = summary(lib.batched_barrett(q, a))
        out[f"mul_barrett_q{q}"] = summary(lib.mul_barrett_lazy(q, a % U(q), b % U(q)))
        if q % 2:
            out[f"mul_hybrid_q{q}"] = summary(lib.mul_hybrid_lazy(q, a % U(2 * q), b % U(2 * q)))
            out[f"montgomery128_q{q}"] = summary(lib.montgomery_128_lazy(q, np.stack([a, b % U(q)], axis=1)))

    # ---- single-limb transforms (tests/ntt_t.cpp grid + config primes) -----------
    grid = [(q, l) for q in P.NTT_TEST_Q for l in (4, 7, 11, 12, 13, 14, 15, 16)]   # (16: the largest degree the reference's 16-bit
    # bit reversal allows, permutation.h:41-55; needs 2^17 | q - 1)
    grid += [(P.P50[1], 16), (P.P40[0], 16)]
    grid += [(q, 14) for q in P.C2_MODULI] + [(q, 15) for q in P.C3_MODULI_EXT] + [(q, 13) for q in P.C5_MODULI_EXT]
    for q, logn in grid:
        if (q - 1) % (2 << logn):
            continue
        x = SplitMix(1 if (q, logn) == (P.C1_Q, 12) else 100 + logn).words(1 << logn, q)
        y = lib.ntt(logn, q, x)
        out[f"ntt_q{q}_logn{logn}"] = summary(y)
        out[f"intt_ntt_q{q}_logn{logn}"] = summary(lib.intt(logn, q, y))
    x = SplitMix(9).words(16, 65537)
    out["ntt_small_full"] = summary(lib.ntt(4, 65537, x))
    out["intt_small_full"] = summary(lib.intt(4, 65537, x))

    # ---- RnsIntVec operators, N = 8 in full and N = 1024 digests ---------------
    for n, moduli in ((8, [17179672577, 17179410433, 17176854529]), (1024, P.P40[:4])):
        rng = SplitMix(n)
        two_q = [2 * m for m in moduli]
        a = rng.poly((len(moduli), n), two_q)
        b = rng.poly((len(moduli), n), two_q)
        out[f"poly_add_n{n}"] = summary(lib.poly_add(moduli, a, b))
        out[f"poly_sub_n{n}"] = summary(lib.poly_sub(moduli, a, b))
        out[f"poly_mul_n{n}"] = summary(lib.poly_mul(moduli, a, b))
        out[f"poly_scalar_mul_n{n}"] = summary(lib.poly_scalar_mul(moduli, a, 2**63 + 12345))
        out[f"poly_rns_scalar_mul_n{n}"] = summary(lib.poly_rns_scalar_mul(moduli, a, [int(w) for w in SplitMix(3).words(len(moduli))]))
        out[f"poly_ntt_n{n}"] = summary(lib.poly_ntt(moduli, a))
        out[f"poly_intt_n{n}"] = summary(lib.poly_intt(moduli, a))
        out[f"poly_involution_n{n}"] = summary(lib.poly_involution(a))
        out[f"poly_cycle3_n{n}"] = summary(lib.poly_cycle(a, 3))

    # ---- key switch / drop-last-prime / scheme level ----------------------------
    scheme = [("n8", 3, [1099510054913, 1073479681, 1072496641, 1099507695617]),
              ("n1024", 10, P.P40[:3] + [P.P50[0]])]
    if big:
        scheme += [("c5", P.C5_LOGN, P.C5_MODULI_EXT), ("c3", P.C3_LOGN, P.C3_MODULI_EXT)]
    for tag, logn, mext in scheme:
        n, L = 1 << logn, len(mext) - 1
        rng = SplitMix({"n8": 8, "n1024": 1024, "c5": 5, "c3": 3}[tag])
        q = mext[:L]
        ct1 = rng.poly((2, L, n), q)
        ct2 = rng.poly((2, L, n), q)
        key = rng.poly((L, 2, L + 1, n), mext)
        quad = lib.mult_low_level(q, ct1, ct2)
        out[f"mult_low_level_{tag}"] = summary(quad)
        ext = lib.ext_prod(mext, quad[2], key)
        out[f"ext_prod_{tag}"] = summary(ext)
        out[f"ckks_rescale_ext_{tag}"] = summary(lib.ckks_rescale(mext, ext))
        out[f"ckks_rescale_ct_{tag}"] = summary(lib.ckks_rescale(q, ct1))
        out[f"bgv_mod_drop_t65537_{tag}"] = summary(lib.bgv_mod_drop(q, 65537, ct2))
        out[f"ckks_relinearize_{tag}"] = summary(lib.ckks_relinearize(mext, quad, key))
        out[f"bgv_relinearize_{tag}"] = summary(lib.bgv_relinearize(mext, quad, key))
        out[f"ckks_rotate3_{tag}"] = summary(lib.ckks_rotate(mext, ct1, key, 3))
        out[f"ckks_conjugate_{tag}"] = summary(lib.ckks_conjugate(mext, ct2, key))
        out[f"ckks_mult_{tag}"] = summary(lib.ckks_mult(mext, ct1, ct2, key))
        out[f"bgv_mult_{tag}"] = summary(lib.bgv_mult(mext, 65537, ct1, ct2, key))
        # ---- either side of the path ------------------------
        noise, c1, pt, sk = P.edge_case(rng, logn, q)
        enc = lib.rlwe_encrypt_core(q, noise, c1, pt, sk)
        out[f"rlwe_encrypt_core_{tag}"] = summary(enc)
        out[f"rlwe_decrypt_core_{tag}"] = summary(lib.rlwe_decrypt_core(q, enc, sk))
        out[f"base_from_single_t65537_{tag}"] = summary(lib.rns_base_from_single(65537, q, rng.words(n, 2 * 65537)))
        out[f"base_from_single_p_{tag}"] = summary(lib.rns_base_from_single(mext[L], q, rng.words(n, 2 * mext[L])))
        out[f"base_to_single_t65537_{tag}"] = summary(lib.rns_base_to_single_small(q, 65537, P.small_rns_poly(rng, n, q))[1])
        out[f"base_to_single_crt_t65537_{tag}"] = summary(lib.rns_base_to_single(q, 65537, rng.poly((L, n), q)))
    return out


def wire_fixture_case():
    """inputs of tests/golden/ckks_mult_n8.hehubamd: the "n8" scheme case above"""
    mext = [1099510054913, 1073479681, 1072496641, 1099507695617]
    n, L = 8, 3
    rng = SplitMix(8)
    ct1 = rng.poly((2, L, n), mext[:L])
    ct2 = rng.poly((2, L, n), mext[:L])
    key = rng.poly((L, 2, L + 1, n), mext)
    return mext, ct1, ct2, key
