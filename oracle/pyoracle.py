"""ctypes front-end for the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Two libraries share one flat-array calling convention (see hehub_oracle.h):

* ``Oracle("orc")``  -> oracle/libhehub_oracle.so  (the C restatement)
* ``Oracle("ref")``  -> oracle/_ref/libhehub_ref.so (the unmodified reference
  compiled by oracle/Makefile; exists only where it was built)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product (hehub_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libhehub_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libhehub_ref.so")

u64 = C.c_uint64
szt = C.c_size_t
P = C.c_void_p


def build(ref: bool = True) -> None:
    """Compile the checker libraries (building the checker is not using it)."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)
        if os.path.exists(os.path.join(HERE, "..", "hehub_amd", "lib", "libhehub_amd.so")):
            # hehub's own test-suite / benchmark program and our end-to-end program over the binding (GPU box only)
            subprocess.run(["make", "-s", "-j8", "-C", HERE, "ref_tests", "ref_e2e", "ref_bench", "ref_chain", "ref_indep", "ref_randprog_amd", "ref_matvec"], check=True)
        subprocess.run(["make", "-s", "-C", HERE, "ref_randprog", "ref_rotbench"], check=True)   # hehub alone: need no engine library


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(P)


def _mods(m) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(m, dtype=np.uint64))


class Oracle:
    def __init__(self, kind: str = "orc"):
        assert kind in ("orc", "ref")
        self.kind = kind
        path = ORACLE_SO if kind == "orc" else REF_SO
        if not os.path.exists(path):
            if kind == "orc":
                build(ref=False)
            else:
                raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self._sig()

    # -- signatures ------------------------------------------------------
    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, f"{self.kind}_{name}")
        fn.restype = restype
        fn.argtypes = argtypes
        return fn

    def _sig(self):
        f = self._f
        self._harvey = f("mul_mod_harvey_lazy", u64, [u64, u64, u64, u64])
        self._inv = f("inverse_mod_prime", u64, [u64, u64])
        self._pow = f("pow_mod", u64, [u64, u64, u64])
        self._root = f("get_2nth_unity_root", C.c_int, [u64, u64, C.POINTER(u64)])
        self._bitrev = f("bit_rev", u64, [u64, C.c_int])
        for nm in ("batched_barrett_lazy", "batched_barrett", "batched_reduce_strict"):
            setattr(self, "_" + nm, f(nm, None, [u64, szt, P]))
        self._hybrid = f("batched_mul_mod_hybrid_lazy", None, [u64, szt, P, P, P])
        self._barrett_mul = f("batched_mul_mod_barrett_lazy", None, [u64, szt, P, P, P])
        self._mont = f("batched_montgomery_128_lazy", None, [u64, szt, P, P])
        self._ntt = f("ntt_negacyclic_inplace_lazy", C.c_int, [szt, u64, P])
        self._intt = f("intt_negacyclic_inplace_lazy", C.c_int, [szt, u64, P])
        self._padd = f("poly_add_inplace", None, [szt, szt, P, P, P])
        self._psub = f("poly_sub_inplace", None, [szt, szt, P, P, P])
        self._pmul = f("poly_mul", None, [szt, szt, P, P, P, P])
        self._psmul = f("poly_scalar_mul_inplace", None, [szt, szt, P, P, u64])
        self._prmul = f("poly_rns_scalar_mul_inplace", None, [szt, szt, P, P, P])
        self._pntt = f("poly_ntt", C.c_int, [szt, szt, P, P])
        self._pintt = f("poly_intt", C.c_int, [szt, szt, P, P])
        self._pstrict = f("poly_reduce_strict", None, [szt, szt, P, P])
        self._pinvol = f("poly_involution", None, [szt, szt, P, P])
        self._pcycle = f("poly_cycle", None, [szt, szt, szt, P, P])
        self._ext = f("ext_prod_montgomery", C.c_int, [szt, szt, P, P, P, P])
        self._rescale = f("ckks_rescale_by_one_prime", C.c_int, [szt, szt, P, P, P])
        self._moddrop = f("bgv_mod_drop_one_prime", C.c_int, [szt, szt, P, u64, P, P])
        self._mll = f("mult_low_level", None, [szt, szt, P, P, P, P])
        self._crelin = f("ckks_relinearize", C.c_int, [szt, szt, P, P, P, P])
        self._brelin = f("bgv_relinearize", C.c_int, [szt, szt, P, u64, P, P, P])
        self._crot = f("ckks_rotate", C.c_int, [szt, szt, P, szt, P, P, P])
        self._cconj = f("ckks_conjugate", C.c_int, [szt, szt, P, P, P, P])
        self._enc = f("rlwe_encrypt_core", C.c_int, [szt, szt, P, P, P, P, P, P])
        self._dec = f("rlwe_decrypt_core", C.c_int, [szt, szt, P, P, P, P])
        self._bfs = f("rns_base_from_single", None, [szt, u64, szt, P, P, P])
        self._bts = f("rns_base_to_single_small", C.c_int, [szt, szt, P, u64, P, P])
        self._bt1 = f("rns_base_to_single", None, [szt, szt, P, u64, P, P])
        self._cmult = f("ckks_mult_relin_rescale", C.c_int, [szt, szt, P, P, P, P, P])
        self._bmult = f("bgv_mult_relin_modswitch", C.c_int, [szt, szt, P, u64, P, P, P, P])
        if self.kind == "orc":
            self._fnv = f("fnv1a64", u64, [P, szt])
            self._fill = f("splitmix_fill", None, [C.POINTER(u64), u64, szt, P])
            self._nttf = f("ntt_factors", C.c_int, [u64, szt, P, P])
            self._inttf = f("intt_factors", C.c_int, [u64, szt, P, P])
            self._mqinv = f("minus_q_inv_mod_2to64", u64, [u64])
            self._r64 = f("2to64_mod_q", u64, [u64])
            self._hq = f("harvey_quotient", u64, [u64, u64])
        else:
            self._tntt = f("time_ntt", C.c_double, [szt, u64, C.c_int, szt, P])
            self._tckks = f("time_ckks_mult", C.c_double, [szt, szt, P, P, P, P, szt])
            self._tbgv = f("time_bgv_mult", C.c_double, [szt, szt, P, u64, P, P, P, szt])

    # -- scalar ----------------------------------------------------------
    def mul_mod_harvey_lazy(self, q, a, b, bh):
        return int(self._harvey(q, a, b, bh))

    def inverse_mod_prime(self, elem, prime):
        return int(self._inv(elem, prime))

    def pow_mod(self, q, base, index):
        return int(self._pow(q, base, index))

    def unity_root(self, q, n):
        r = u64(0)
        rc = self._root(q, n, C.byref(r))
        if rc != 0:
            raise ValueError("2N doesn't divide (modulus - 1)")
        return int(r.value)

    def bit_rev(self, x, bits):
        return int(self._bitrev(x, bits))

    # -- batched ---------------------------------------------------------
    def batched_barrett_lazy(self, q, v):
        v = v.copy(); self._batched_barrett_lazy(q, v.size, _p(v)); return v

    def batched_barrett(self, q, v):
        v = v.copy(); self._batched_barrett(q, v.size, _p(v)); return v

    def batched_reduce_strict(self, q, v):
        v = v.copy(); self._batched_reduce_strict(q, v.size, _p(v)); return v

    def mul_hybrid_lazy(self, q, a, b):
        out = np.empty_like(a); self._hybrid(q, a.size, _p(a), _p(b), _p(out)); return out

    def mul_barrett_lazy(self, q, a, b):
        out = np.empty_like(a); self._barrett_mul(q, a.size, _p(a), _p(b), _p(out)); return out

    def montgomery_128_lazy(self, q, in128):
        """in128: uint64[n,2] = {lo,hi}."""
        n = in128.shape[0]
        out = np.empty(n, dtype=np.uint64)
        self._mont(q, n, _p(np.ascontiguousarray(in128)), _p(out))
        return out

    # -- transforms ------------------------------------------------------
    def ntt(self, logn, q, x):
        x = x.copy()
        rc = self._ntt(logn, q, _p(x))
        if rc != 0:
            raise ValueError(f"ntt rc={rc}")
        return x

    def intt(self, logn, q, x):
        x = x.copy()
        rc = self._intt(logn, q, _p(x))
        if rc != 0:
            raise ValueError(f"intt rc={rc}")
        return x

    # -- polynomial level (arrays [L,N]) ----------------------------------
    def poly_add(self, moduli, a, b):
        a = a.copy(); L, n = a.shape; self._padd(n, L, _p(_mods(moduli)), _p(a), _p(b)); return a

    def poly_sub(self, moduli, a, b):
        a = a.copy(); L, n = a.shape; self._psub(n, L, _p(_mods(moduli)), _p(a), _p(b)); return a

    def poly_mul(self, moduli, a, b):
        L, n = a.shape; out = np.empty_like(a)
        self._pmul(n, L, _p(_mods(moduli)), _p(a), _p(b), _p(out)); return out

    def poly_scalar_mul(self, moduli, a, scalar):
        a = a.copy(); L, n = a.shape; self._psmul(n, L, _p(_mods(moduli)), _p(a), scalar); return a

    def poly_rns_scalar_mul(self, moduli, a, scalars):
        a = a.copy(); L, n = a.shape
        self._prmul(n, L, _p(_mods(moduli)), _p(a), _p(_mods(scalars))); return a

    def poly_ntt(self, moduli, a):
        a = a.copy(); L, n = a.shape
        rc = self._pntt(n.bit_length() - 1, L, _p(_mods(moduli)), _p(a))
        if rc != 0:
            raise ValueError(f"poly_ntt rc={rc}")
        return a

    def poly_intt(self, moduli, a):
        a = a.copy(); L, n = a.shape
        rc = self._pintt(n.bit_length() - 1, L, _p(_mods(moduli)), _p(a))
        if rc != 0:
            raise ValueError(f"poly_intt rc={rc}")
        return a

    def poly_reduce_strict(self, moduli, a):
        a = a.copy(); L, n = a.shape; self._pstrict(n, L, _p(_mods(moduli)), _p(a)); return a

    def poly_involution(self, a):
        L, n = a.shape; out = np.empty_like(a)
        self._pinvol(n.bit_length() - 1, L, _p(a), _p(out)); return out

    def poly_cycle(self, a, step):
        L, n = a.shape; out = np.empty_like(a)
        self._pcycle(n.bit_length() - 1, L, step, _p(a), _p(out)); return out

    # -- key switch / scheme level ----------------------------------------
    def ext_prod(self, moduli_ext, pt, key):
        L, n = pt.shape
        out = np.empty((2, L + 1, n), dtype=np.uint64)
        rc = self._ext(n.bit_length() - 1, L, _p(_mods(moduli_ext)), _p(pt), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"ext_prod rc={rc}")
        return out

    def ckks_rescale(self, moduli, ct):
        _, L, n = ct.shape
        out = np.empty((2, L - 1, n), dtype=np.uint64)
        rc = self._rescale(n.bit_length() - 1, L, _p(_mods(moduli)), _p(ct), _p(out))
        if rc != 0:
            raise ValueError(f"rescale rc={rc}")
        return out

    def bgv_mod_drop(self, moduli, t, ct):
        _, L, n = ct.shape
        out = np.empty((2, L - 1, n), dtype=np.uint64)
        rc = self._moddrop(n.bit_length() - 1, L, _p(_mods(moduli)), t, _p(ct), _p(out))
        if rc != 0:
            raise ValueError(f"mod_drop rc={rc}")
        return out

    def mult_low_level(self, moduli, ct1, ct2):
        _, L, n = ct1.shape
        out = np.empty((3, L, n), dtype=np.uint64)
        self._mll(n, L, _p(_mods(moduli)), _p(ct1), _p(ct2), _p(out)); return out

    def ckks_relinearize(self, moduli_ext, quad, key):
        _, L, n = quad.shape
        out = np.empty((2, L, n), dtype=np.uint64)
        rc = self._crelin(n.bit_length() - 1, L, _p(_mods(moduli_ext)), _p(quad), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"relin rc={rc}")
        return out

    def bgv_relinearize(self, moduli_ext, quad, key, inner_t=1):
        _, L, n = quad.shape
        out = np.empty((2, L, n), dtype=np.uint64)
        rc = self._brelin(n.bit_length() - 1, L, _p(_mods(moduli_ext)), inner_t, _p(quad), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"relin rc={rc}")
        return out

    def ckks_rotate(self, moduli_ext, ct, key, step):
        _, L, n = ct.shape
        out = np.empty((2, L, n), dtype=np.uint64)
        rc = self._crot(n.bit_length() - 1, L, _p(_mods(moduli_ext)), step, _p(ct), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"rotate rc={rc}")
        return out

    def ckks_conjugate(self, moduli_ext, ct, key):
        _, L, n = ct.shape
        out = np.empty((2, L, n), dtype=np.uint64)
        rc = self._cconj(n.bit_length() - 1, L, _p(_mods(moduli_ext)), _p(ct), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"conjugate rc={rc}")
        return out

    def rlwe_encrypt_core(self, moduli, noise, c1, pt, sk):
        L, n = c1.shape
        assert noise.dtype == np.int64 and noise.flags["C_CONTIGUOUS"]
        ct = np.empty((2, L, n), dtype=np.uint64)
        rc = self._enc(n.bit_length() - 1, L, _p(_mods(moduli)), noise.ctypes.data_as(P), _p(c1), _p(pt), _p(sk), _p(ct))
        if rc != 0:
            raise ValueError(f"encrypt_core rc={rc}")
        return ct

    def rlwe_decrypt_core(self, moduli, ct, sk):
        _, L, n = ct.shape
        pt = np.empty((L, n), dtype=np.uint64)
        rc = self._dec(n.bit_length() - 1, L, _p(_mods(moduli)), _p(ct), _p(sk), _p(pt))
        if rc != 0:
            raise ValueError(f"decrypt_core rc={rc}")
        return pt

    def rns_base_from_single(self, old_modulus, new_moduli, x):
        n, L = x.size, len(new_moduli)
        out = np.empty((L, n), dtype=np.uint64)
        self._bfs(n, old_modulus, L, _p(_mods(new_moduli)), _p(x), _p(out))
        return out

    def rns_base_to_single_small(self, old_moduli, new_modulus, x):
        """returns (all_small, out[n]); the compiled reference always reports 1 (it falls into its CRT branch itself)."""
        L, n = x.shape
        out = np.empty(n, dtype=np.uint64)
        ok = self._bts(n, L, _p(_mods(old_moduli)), new_modulus, _p(x), _p(out))
        return bool(ok), out

    def rns_base_to_single(self, old_moduli, new_modulus, x):
        """rns_base_transform(poly, {new_modulus}): small-coefficient branch or CRT composition, as the reference decides."""
        L, n = x.shape
        out = np.empty(n, dtype=np.uint64)
        self._bt1(n, L, _p(_mods(old_moduli)), new_modulus, _p(x), _p(out))
        return out

    def ckks_mult(self, moduli_ext, ct1, ct2, key):
        _, L, n = ct1.shape
        out = np.empty((2, L - 1, n), dtype=np.uint64)
        rc = self._cmult(n.bit_length() - 1, L, _p(_mods(moduli_ext)), _p(ct1), _p(ct2), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"ckks_mult rc={rc}")
        return out

    def bgv_mult(self, moduli_ext, t, ct1, ct2, key):
        _, L, n = ct1.shape
        out = np.empty((2, L - 1, n), dtype=np.uint64)
        rc = self._bmult(n.bit_length() - 1, L, _p(_mods(moduli_ext)), t, _p(ct1), _p(ct2), _p(key), _p(out))
        if rc != 0:
            raise ValueError(f"bgv_mult rc={rc}")
        return out

    # -- oracle-only helpers ----------------------------------------------
    def fnv(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        return int(self._fnv(a.ctypes.data_as(P), a.nbytes))

    def ntt_factors(self, q, logn):
        n = 1 << logn
        s = np.empty(n, dtype=np.uint64); h = np.empty(n, dtype=np.uint64)
        rc = self._nttf(q, logn, _p(s), _p(h))
        if rc != 0:
            raise ValueError(f"ntt_factors rc={rc}")
        return s, h

    def intt_factors(self, q, logn):
        n = 1 << logn
        s = np.empty(2 * n, dtype=np.uint64); h = np.empty(2 * n, dtype=np.uint64)
        rc = self._inttf(q, logn, _p(s), _p(h))
        if rc != 0:
            raise ValueError(f"intt_factors rc={rc}")
        return s, h

    # -- reference-only timing --------------------------------------------
    def time_ntt(self, logn, q, inverse, iters, x):
        x = x.copy()
        return float(self._tntt(logn, q, int(inverse), iters, _p(x)))

    def time_ckks_mult(self, moduli_ext, ct1, ct2, key, iters):
        _, L, n = ct1.shape
        return float(self._tckks(n.bit_length() - 1, L, _p(_mods(moduli_ext)), _p(ct1), _p(ct2), _p(key), iters))

    def time_bgv_mult(self, moduli_ext, t, ct1, ct2, key, iters):
        _, L, n = ct1.shape
        return float(self._tbgv(n.bit_length() - 1, L, _p(_mods(moduli_ext)), t, _p(ct1), _p(ct2), _p(key), iters))


class SplitMix:
    """splitmix64 stream -- the input generator shared by
    fixtures, parity tests and bench."""

    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def words(self, n: int, q: int = 0) -> np.ndarray:
        # vectorised: state_i = seed + (i+1)*gamma
        gamma = np.uint64(0x9E3779B97F4A7C15)
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = np.uint64(self.state) + idx * gamma
            self.state = int(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        if q:
            z = z % np.uint64(q)
        return z

    def poly(self, shape, moduli) -> np.ndarray:
        """uniform words in [0,q_k) for an array [..., L, N]; one stream, limb-major."""
        shape = tuple(shape)
        L, n = shape[-2], shape[-1]
        out = np.empty(shape, dtype=np.uint64)
        flat = out.reshape(-1, L, n)
        for b in range(flat.shape[0]):
            for k in range(L):
                flat[b, k] = self.words(n, int(moduli[k]))
        return out
