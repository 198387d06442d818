"""Python front-end of the C ABI for tests and benchmarks.

PyTorch is plumbing here: it owns device buffers (int64 tensors holding the raw
u64 words) and the current HIP stream, and provides torch.distributed for
multi-GPU runs.  Every operation is a call into libhehub_amd.so; nothing is
computed by torch or numpy.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import capi


class HpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[hp_status {code}] {msg}")
        self.code = code
        self.msg = msg


class InvalidArgument(HpError, ValueError):
    """The reference throws std::invalid_argument for the same input."""


def _u64arr(v: Sequence[int]):
    return (capi.u64 * len(v))(*[int(x) for x in v])


class Engine:
    """One engine context on one GPU (include/hehub_amd.h: hp_ctx)."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch  # noqa: F401  (loads the HIP runtime the library binds to)

        self.torch = torch
        if not torch.cuda.is_available():
            raise capi.EngineMissing("no HIP device visible: the engine needs an MI355X (no CPU fallback)")
        self.lib = capi.load()
        self.device = device
        torch.cuda.set_device(device)
        h = capi.P()
        rc = self.lib.hp_ctx_create(device, C.byref(h))
        if rc != capi.HP_OK:
            raise HpError(rc, "hp_ctx_create failed")
        self.h = h
        if use_torch_stream:
            self.use_stream(torch.cuda.current_stream(device))

    def fork(self) -> "Engine":
        """A second lane on the same GPU (hp_ctx_fork): its own stream and scratch workspace, the family's tables and lock; calls on it
        overlap on the device with calls on this one.  Order them with wait_for where one reads what the other wrote.
        A lane's stream is not torch's: torch's caching allocator recycles a freed tensor's block at once, so keep every tensor a lane's
        calls touch referenced until that lane has been synchronised (sync / wait_for), and wait for copies torch makes on its own stream
        before a lane reads them."""
        e = object.__new__(Engine)
        e.torch, e.lib, e.device = self.torch, self.lib, self.device
        h = capi.P()
        self._chk(self.lib.hp_ctx_fork(self.h, C.byref(h)))
        e.h = h
        return e

    def wait_for(self, other: "Engine"):
        """everything enqueued on this context from now on runs after everything `other` has enqueued so far (device-side event)"""
        self._chk(self.lib.hp_ctx_wait_for(self.h, other.h))

    def gather_rows(self, rows, words: int, out):
        """rows: device tensors (each `words` int64 words, anywhere) -> out[len(rows)][words] by one kernel per 64 rows"""
        ptrs = (capi.P * len(rows))(*[r.data_ptr() for r in rows])
        self._chk(self.lib.hp_dev_gather_rows(self.h, len(rows), words, ptrs, self._ptr(out)))
        return out

    def scatter_rows(self, packed, words: int, rows):
        ptrs = (capi.P * len(rows))(*[r.data_ptr() for r in rows])
        self._chk(self.lib.hp_dev_scatter_rows(self.h, len(rows), words, self._ptr(packed), ptrs))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------
    def _chk(self, rc: int):
        if rc == capi.HP_OK:
            return
        msg = self.lib.hp_last_error(self.h).decode()
        if rc == capi.HP_EINVAL:
            raise InvalidArgument(rc, msg)
        raise HpError(rc, msg)

    def use_stream(self, stream):
        self._chk(self.lib.hp_ctx_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def sync(self):
        self._chk(self.lib.hp_sync(self.h))

    def workspace_bytes(self) -> int:
        return int(self.lib.hp_ctx_workspace_bytes(self.h))

    def release_workspace(self):
        self._chk(self.lib.hp_ctx_release_workspace(self.h))

    def force_generic(self, on: bool):
        self._chk(self.lib.hp_ctx_set_force_generic(self.h, int(on)))

    def set_parity_level(self, level: str):
        """"B" (default): raw words identical to hehub's.  "A": the scheme-level pipelines return canonical residues through
        the FP64 transforms (include/hehub_amd.h: hp_ctx_set_parity_level); the NTT / mod-arith primitives are never affected."""
        self._chk(self.lib.hp_ctx_set_parity_level(self.h, {"B": 0, "A": 1}[level.upper()]))

    def parity_level(self) -> str:
        return "BA"[self.lib.hp_ctx_get_parity_level(self.h)]

    def to_device(self, a: np.ndarray):
        assert a.dtype == np.uint64
        return self.torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(f"cuda:{self.device}")

    @staticmethod
    def to_host(t) -> np.ndarray:
        return t.detach().cpu().numpy().view(np.uint64)

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.int64, device=f"cuda:{self.device}")

    @staticmethod
    def _ptr(t):
        assert t.is_contiguous() and t.element_size() == 8
        return C.c_void_p(t.data_ptr())

    # -- profiling ---------------------------------------------------------
    def prof_begin(self, family: str):
        self._chk(self.lib.hp_prof_begin(self.h, family.encode()))

    def prof_end(self):
        n = capi.szt(0)
        ms = C.c_double(0)
        self._chk(self.lib.hp_prof_end(self.h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def prof_end_families(self):
        """after prof_begin("*"): {family: (launches, total ms)} of every bracketed launch, in order of first appearance"""
        cap = 32
        names = (C.c_char_p * cap)()
        launches = (capi.szt * cap)()
        ms = (C.c_double * cap)()
        count = capi.szt(0)
        self._chk(self.lib.hp_prof_end_families(self.h, cap, names, launches, ms, C.byref(count)))
        return {names[i].decode(): (int(launches[i]), float(ms[i])) for i in range(int(count.value))}

    # -- drop-in host calls (numpy in, numpy out) ----------------------------
    def host_ntt(self, logn: int, q: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x.copy())
        self._chk(self.lib.hp_ntt_negacyclic_inplace_lazy(self.h, logn, q, x.ctypes.data_as(capi.P)))
        return x

    def host_intt(self, logn: int, q: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x.copy())
        self._chk(self.lib.hp_intt_negacyclic_inplace_lazy(self.h, logn, q, x.ctypes.data_as(capi.P)))
        return x

    def cache_ntt_factors_strict(self, logn: int, moduli):
        self._chk(self.lib.hp_cache_ntt_factors_strict(self.h, logn, _u64arr(moduli), len(moduli)))

    def _host_inplace(self, fn, q, v):
        v = np.ascontiguousarray(v.copy())
        self._chk(fn(self.h, q, v.size, v.ctypes.data_as(capi.P)))
        return v

    def host_barrett_lazy(self, q, v):
        return self._host_inplace(self.lib.hp_batched_barrett_lazy, q, v)

    def host_barrett(self, q, v):
        return self._host_inplace(self.lib.hp_batched_barrett, q, v)

    def host_reduce_strict(self, q, v):
        return self._host_inplace(self.lib.hp_batched_reduce_strict, q, v)

    def host_mul_hybrid_lazy(self, q, a, b):
        out = np.empty_like(a)
        self._chk(self.lib.hp_batched_mul_mod_hybrid_lazy(self.h, q, a.size, a.ctypes.data_as(capi.P),
                                                          b.ctypes.data_as(capi.P), out.ctypes.data_as(capi.P)))
        return out

    def host_mul_barrett_lazy(self, q, a, b):
        out = np.empty_like(a)
        self._chk(self.lib.hp_batched_mul_mod_barrett_lazy(self.h, q, a.size, a.ctypes.data_as(capi.P),
                                                           b.ctypes.data_as(capi.P), out.ctypes.data_as(capi.P)))
        return out

    def host_montgomery_128_lazy(self, q, in128):
        in128 = np.ascontiguousarray(in128)
        out = np.empty(in128.shape[0], dtype=np.uint64)
        self._chk(self.lib.hp_batched_montgomery_128_lazy(self.h, q, out.size, in128.ctypes.data_as(capi.P),
                                                          out.ctypes.data_as(capi.P)))
        return out

    # -- device batches: tensors [batch, L, N] of int64 (raw u64 words) --------
    def ntt_(self, moduli, x):
        B, L, n = x.shape
        self._chk(self.lib.hp_dev_ntt(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(x)))
        return x

    def intt_(self, moduli, x, strict: bool = False):
        B, L, n = x.shape
        self._chk(self.lib.hp_dev_intt(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(x), int(strict)))
        return x

    def ntt_residues_(self, moduli, x):
        """forward transforms in place to CANONICAL residues (hp_dev_ntt_residues: the FP64 kernels; == oracle ntt words mod q)"""
        B, L, n = x.shape
        self._chk(self.lib.hp_dev_ntt_residues(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(x)))
        return x

    def intt_residues_(self, moduli, x):
        """inverse transforms in place, strict (hp_dev_intt_residues: the words of intt_negacyclic_inplace)"""
        B, L, n = x.shape
        self._chk(self.lib.hp_dev_intt_residues(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(x)))
        return x

    def _binary(self, fn, moduli, a, b, out=None):
        B, L, n = a.shape
        out = self.empty(a.shape) if out is None else out
        self._chk(fn(self.h, n, L, _u64arr(moduli), B, self._ptr(a), self._ptr(b), self._ptr(out)))
        return out

    def poly_add(self, moduli, a, b, out=None):
        return self._binary(self.lib.hp_dev_poly_add, moduli, a, b, out)

    def poly_fold_rows(self, moduli, chains, negate, out=None):
        """chains[p] = the term polynomials (device tensors [L][n], anywhere) of polynomial p; out[p] = ((x0 op1 x1) op2 x2) ... with
        op_j = -= where negate[j] (include/hehub_amd.h: hp_dev_poly_fold_rows): the words of the chain of += / -= calls"""
        polys, terms = len(chains), len(chains[0])
        L, n = chains[0][0].shape
        if out is None:
            out = self.empty((polys, L, n))
        ptrs = (capi.P * (polys * terms))(*[t.data_ptr() for c in chains for t in c])
        neg = (C.c_uint8 * terms)(*[1 if x else 0 for x in negate])
        self._chk(self.lib.hp_dev_poly_fold_rows(self.h, n, L, _u64arr(moduli), polys, terms, neg, ptrs, self._ptr(out)))
        return out

    def poly_sub(self, moduli, a, b, out=None):
        return self._binary(self.lib.hp_dev_poly_sub, moduli, a, b, out)

    def poly_mul(self, moduli, a, b, out=None):
        return self._binary(self.lib.hp_dev_poly_mul, moduli, a, b, out)

    def poly_scalar_mul(self, moduli, a, scalars, out=None):
        B, L, n = a.shape
        if isinstance(scalars, int):
            scalars = [scalars] * L
        if len(scalars) != L:
            raise InvalidArgument(capi.HP_EINVAL, "Numbers of RNS component mismatch.")
        out = self.empty(a.shape) if out is None else out
        self._chk(self.lib.hp_dev_poly_scalar_mul(self.h, n, L, _u64arr(moduli), B, _u64arr(scalars), self._ptr(a),
                                                  self._ptr(out)))
        return out

    def poly_reduce_strict_(self, moduli, x):
        B, L, n = x.shape
        self._chk(self.lib.hp_dev_poly_reduce_strict(self.h, n, L, _u64arr(moduli), B, self._ptr(x)))
        return x

    def copy(self, src, out=None):
        """deep copy of device words (hp_dev_copy)"""
        out = self.empty(src.shape) if out is None else out
        # the kernel writes src.numel() words at out's address: a short, strided or foreign `out` would be written past its end
        for t in (src, out):
            if t.element_size() != 8 or not t.is_contiguous() or t.device.type != "cuda" or t.device.index != self.device:
                raise InvalidArgument(capi.HP_EINVAL, "copy: contiguous 8-byte tensors on the engine's device")
        if out.numel() != src.numel():
            raise InvalidArgument(capi.HP_EINVAL, f"copy: out holds {out.numel()} words, src {src.numel()}")
        self._chk(self.lib.hp_dev_copy(self.h, src.numel(), self._ptr(src), self._ptr(out)))
        return out

    def poly_involution(self, a):
        B, L, n = a.shape
        out = self.empty(a.shape)
        self._chk(self.lib.hp_dev_poly_involution(self.h, n.bit_length() - 1, L, B, self._ptr(a), self._ptr(out)))
        return out

    def poly_cycle(self, a, step: int):
        B, L, n = a.shape
        out = self.empty(a.shape)
        self._chk(self.lib.hp_dev_poly_cycle(self.h, n.bit_length() - 1, L, B, step, self._ptr(a), self._ptr(out)))
        return out

    # -- scheme level: ciphertext batches [batch, 2|3, L, N] -------------------
    def mult_low_level(self, moduli, ct1, ct2):
        B, two, L, n = ct1.shape
        out = self.empty((B, 3, L, n))
        self._chk(self.lib.hp_dev_mult_low_level(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(ct1),
                                                 self._ptr(ct2), self._ptr(out)))
        return out

    def ext_prod(self, moduli_ext, pt, key):
        B, L, n = pt.shape
        out = self.empty((B, 2, L + 1, n))
        self._chk(self.lib.hp_dev_ext_prod_montgomery(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B,
                                                      self._ptr(pt), self._ptr(key), self._ptr(out)))
        return out

    def ckks_rescale(self, moduli, ct):
        B, two, L, n = ct.shape
        out = self.empty((B, 2, max(L - 1, 1), n))
        self._chk(self.lib.hp_dev_ckks_rescale(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(ct),
                                               self._ptr(out)))
        return out

    def bgv_mod_switch(self, moduli, t, ct):
        B, two, L, n = ct.shape
        out = self.empty((B, 2, max(L - 1, 1), n))
        self._chk(self.lib.hp_dev_bgv_mod_switch(self.h, n.bit_length() - 1, L, _u64arr(moduli), t, B, self._ptr(ct),
                                                 self._ptr(out)))
        return out

    def ckks_relinearize(self, moduli_ext, quad, key):
        B, three, L, n = quad.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_relinearize(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B,
                                                   self._ptr(quad), self._ptr(key), self._ptr(out)))
        return out

    def bgv_relinearize(self, moduli_ext, quad, key, inner_t: int = 1):
        B, three, L, n = quad.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_bgv_relinearize(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), inner_t, B,
                                                  self._ptr(quad), self._ptr(key), self._ptr(out)))
        return out

    def ckks_rotate(self, moduli_ext, ct, key, step: int):
        B, two, L, n = ct.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_rotate(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B, step, self._ptr(ct),
                                              self._ptr(key), self._ptr(out)))
        return out

    def ckks_conjugate(self, moduli_ext, ct, key):
        B, two, L, n = ct.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_conjugate(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B, self._ptr(ct),
                                                 self._ptr(key), self._ptr(out)))
        return out

    # -- extensions beyond the reference (include/hehub_amd.h) ----------------
    def ckks_mult_at(self, moduli_ext, key_L0: int, ct1, ct2, key):
        """ckks::mult + rescale at level L = ct limbs with a key generated for key_L0 >= L ciphertext moduli."""
        B, _, L, n = ct1.shape
        out = self.empty((B, 2, L - 1, n))
        self._chk(self.lib.hp_dev_ckks_mult_relin_rescale_at(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B,
                                                             self._ptr(ct1), self._ptr(ct2), self._ptr(key), self._ptr(out)))
        return out

    def ckks_rotate_at(self, moduli_ext, key_L0: int, ct, key, step: int):
        B, _, L, n = ct.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_rotate_at(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B, step,
                                                 self._ptr(ct), self._ptr(key), self._ptr(out)))
        return out

    def ckks_mult_rows(self, moduli_ext, key_L0: int, pairs, key):
        """ckks::mult + rescale with the operands by address: pairs[b] = (a0, a1, b0, b1), device tensors [L][n] anywhere"""
        B = len(pairs)
        L, n = pairs[0][0].shape
        out = self.empty((B, 2, L - 1, n))
        pp = (capi.P * (4 * B))(*[p.data_ptr() for quad in pairs for p in quad])
        self._chk(self.lib.hp_dev_ckks_mult_relin_rescale_rows(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B, pp,
                                                               self._ptr(key), self._ptr(out)))
        return out

    def bgv_mult_rows(self, moduli_ext, t: int, pairs, key):
        B = len(pairs)
        L, n = pairs[0][0].shape
        out = self.empty((B, 2, L - 1, n))
        pp = (capi.P * (4 * B))(*[p.data_ptr() for quad in pairs for p in quad])
        self._chk(self.lib.hp_dev_bgv_mult_relin_modswitch_rows(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), t, B, pp,
                                                                self._ptr(key), self._ptr(out)))
        return out

    def ckks_rotate_many(self, moduli_ext, key_L0: int, ct, keys, steps, conj=None):
        """ciphertext b rotated by steps[b] (conjugated where conj[b]) and switched with ITS OWN key keys[b] (device tensors)."""
        B, _, L, n = ct.shape
        assert len(keys) == B and len(steps) == B
        out = self.empty((B, 2, L, n))
        kp = (capi.P * B)(*[k.data_ptr() for k in keys])
        st = (C.c_size_t * B)(*[int(s) for s in steps])
        cj = (C.c_ubyte * B)(*[1 if c else 0 for c in conj]) if conj is not None else None
        self._chk(self.lib.hp_dev_ckks_rotate_many(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B, st, cj,
                                                   self._ptr(ct), kp, self._ptr(out)))
        return out

    def ckks_rotate_many_rows(self, moduli_ext, key_L0: int, polys, keys, steps, conj=None):
        """the same with the polynomials of ciphertext b anywhere on the device: polys[b] = (poly0, poly1), tensors [L][n]"""
        B = len(polys)
        L, n = polys[0][0].shape
        assert len(keys) == B and len(steps) == B
        out = self.empty((B, 2, L, n))
        pp = (capi.P * (2 * B))(*[p.data_ptr() for pair in polys for p in pair])
        kp = (capi.P * B)(*[k.data_ptr() for k in keys])
        st = (C.c_size_t * B)(*[int(s) for s in steps])
        cj = (C.c_ubyte * B)(*[1 if c else 0 for c in conj]) if conj is not None else None
        self._chk(self.lib.hp_dev_ckks_rotate_many_rows(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B, st, cj, pp, kp,
                                                        self._ptr(out)))
        return out

    def ext_prod_at(self, moduli_ext, key_L0: int, pt, key):
        B, L, n = pt.shape
        out = self.empty((B, 2, L + 1, n))
        self._chk(self.lib.hp_dev_ext_prod_montgomery_at(self.h, n.bit_length() - 1, L, key_L0, _u64arr(moduli_ext), B,
                                                         self._ptr(pt), self._ptr(key), self._ptr(out)))
        return out

    def rns_base_many_to_many(self, old_moduli, new_moduli, x):
        B, L, n = x.shape
        out = self.empty((B, len(new_moduli), n))
        self._chk(self.lib.hp_dev_rns_base_many_to_many(self.h, n, L, _u64arr(old_moduli), len(new_moduli), _u64arr(new_moduli), B,
                                                        self._ptr(x), self._ptr(out)))
        return out

    def hks_switch(self, moduli_ext, k: int, alpha: int, pt, key):
        """hybrid key switch (extension): pt [B][L][n] NTT form, key [dnum][2][L+k][n] -> [B][2][L][n]."""
        B, L, n = pt.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_hks_switch(self.h, n.bit_length() - 1, L, k, alpha, _u64arr(moduli_ext), B, self._ptr(pt),
                                             self._ptr(key), self._ptr(out)))
        return out

    def ckks_rotate_hks(self, moduli_ext, k: int, alpha: int, ct, key, step: int):
        B, _, L, n = ct.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_rotate_hks(self.h, n.bit_length() - 1, L, k, alpha, _u64arr(moduli_ext), B, step,
                                                  self._ptr(ct), self._ptr(key), self._ptr(out)))
        return out

    def ckks_conjugate_hks(self, moduli_ext, k: int, alpha: int, ct, key):
        B, _, L, n = ct.shape
        out = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_ckks_conjugate_hks(self.h, n.bit_length() - 1, L, k, alpha, _u64arr(moduli_ext), B,
                                                     self._ptr(ct), self._ptr(key), self._ptr(out)))
        return out

    def ckks_mult_hks(self, moduli_ext, k: int, alpha: int, ct1, ct2, key, out=None):
        B, _, L, n = ct1.shape
        out = self.empty((B, 2, L - 1, n)) if out is None else out
        self._chk(self.lib.hp_dev_ckks_mult_relin_rescale_hks(self.h, n.bit_length() - 1, L, k, alpha, _u64arr(moduli_ext), B,
                                                              self._ptr(ct1), self._ptr(ct2), self._ptr(key), self._ptr(out)))
        return out

    def ckks_rescale_n(self, moduli, ct, drops: int):
        B, _, L, n = ct.shape
        out = self.empty((B, 2, L - drops, n))
        tmp = self.empty((2, B, 2, L - 1, n)) if drops > 1 else None
        self._chk(self.lib.hp_dev_ckks_rescale_n(self.h, n.bit_length() - 1, L, _u64arr(moduli), drops, B, self._ptr(ct),
                                                 self._ptr(tmp) if tmp is not None else None, self._ptr(out)))
        return out

    # -- either side of the path -------------------------------------------
    def rlwe_encrypt_core(self, moduli, noise, c1, pt, sk):
        B, L, n = c1.shape
        ct = self.empty((B, 2, L, n))
        self._chk(self.lib.hp_dev_rlwe_encrypt_core(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(noise),
                                                    self._ptr(c1), self._ptr(pt), self._ptr(sk), self._ptr(ct)))
        return ct

    def rlwe_decrypt_core(self, moduli, ct, sk):
        B, _, L, n = ct.shape
        pt = self.empty((B, L, n))
        self._chk(self.lib.hp_dev_rlwe_decrypt_core(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, self._ptr(ct),
                                                    self._ptr(sk), self._ptr(pt)))
        return pt

    def rns_base_from_single(self, old_modulus, new_moduli, x):
        B, n = x.shape
        L = len(new_moduli)
        out = self.empty((B, L, n))
        self._chk(self.lib.hp_dev_rns_base_from_single(self.h, n, old_modulus, L, _u64arr(new_moduli), B, self._ptr(x),
                                                       self._ptr(out)))
        return out

    def rns_base_to_single_small(self, old_moduli, new_modulus, x):
        """returns (out[B][n], not_small[B]); not_small != 0 marks polynomials that need the host CRT branch."""
        B, L, n = x.shape
        out = self.empty((B, n))
        flags = self.torch.empty((B,), dtype=self.torch.int32, device=x.device)
        self._chk(self.lib.hp_dev_rns_base_to_single_small(self.h, n, L, _u64arr(old_moduli), new_modulus, B, self._ptr(x),
                                                           self._ptr(out), C.c_void_p(flags.data_ptr())))
        return out, flags

    def rns_base_to_single(self, old_moduli, new_modulus, x):
        """rns_base_transform(poly, {new_modulus}) complete: small-coefficient or CRT branch per polynomial."""
        B, L, n = x.shape
        out = self.empty((B, n))
        self._chk(self.lib.hp_dev_rns_base_to_single(self.h, n, L, _u64arr(old_moduli), new_modulus, B, self._ptr(x),
                                                     self._ptr(out)))
        return out

    def ckks_mult(self, moduli_ext, ct1, ct2, key, out=None):
        B, two, L, n = ct1.shape
        out = self.empty((B, 2, L - 1, n)) if out is None else out
        self._chk(self.lib.hp_dev_ckks_mult_relin_rescale(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B,
                                                          self._ptr(ct1), self._ptr(ct2), self._ptr(key),
                                                          self._ptr(out)))
        return out

    def bgv_mult(self, moduli_ext, t, ct1, ct2, key, out=None, inner_t: bool = False):
        """inner_t: the extension that keeps the key-switched term (hp_dev_bgv_mult_relin_modswitch_t)"""
        B, two, L, n = ct1.shape
        out = self.empty((B, 2, L - 1, n)) if out is None else out
        fn = self.lib.hp_dev_bgv_mult_relin_modswitch_t if inner_t else self.lib.hp_dev_bgv_mult_relin_modswitch
        self._chk(fn(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), t, B,
                                                           self._ptr(ct1), self._ptr(ct2), self._ptr(key),
                                                           self._ptr(out)))
        return out
