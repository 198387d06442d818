#!/usr/bin/env python3
"""Evaluate tests/golden/cases.py through the HIP engine in a process that never loads a checker library.

    python tests/golden/engine_cases.py > results.json

Inputs: splitmix64 streams (tests/splitmix.py, plain numpy).  Digest: the product's hp_wire_fnv1a64.  The JSON also
reports which of the engine / checker shared libraries the process had mapped when it finished
(tests/test_gpu_golden.py asserts: the engine yes, the checkers no).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


class EngineLib:
    """The method names of tests/golden/cases.py over hehub_amd.engine.Engine (batch of one per call)."""

    def __init__(self, eng):
        self.e = eng

    def fnv(self, flat: np.ndarray) -> int:
        flat = np.ascontiguousarray(flat)
        return int(self.e.lib.hp_wire_fnv1a64(flat.ctypes.data_as(C.c_void_p), flat.nbytes))

    # host entry points
    def batched_barrett_lazy(self, q, a): return self.e.host_barrett_lazy(q, a)
    def batched_barrett(self, q, a): return self.e.host_barrett(q, a)
    def mul_barrett_lazy(self, q, a, b): return self.e.host_mul_barrett_lazy(q, np.ascontiguousarray(a), np.ascontiguousarray(b))
    def mul_hybrid_lazy(self, q, a, b): return self.e.host_mul_hybrid_lazy(q, np.ascontiguousarray(a), np.ascontiguousarray(b))
    def montgomery_128_lazy(self, q, in128): return self.e.host_montgomery_128_lazy(q, in128)
    def ntt(self, logn, q, x): return self.e.host_ntt(logn, q, x)
    def intt(self, logn, q, x): return self.e.host_intt(logn, q, x)

    # device batches of one
    def _d(self, a): return self.e.to_device(np.ascontiguousarray(a)[None])
    def _h(self, t): return self.e.to_host(t)[0]

    def poly_add(self, m, a, b): return self._h(self.e.poly_add(m, self._d(a), self._d(b)))
    def poly_sub(self, m, a, b): return self._h(self.e.poly_sub(m, self._d(a), self._d(b)))
    def poly_mul(self, m, a, b): return self._h(self.e.poly_mul(m, self._d(a), self._d(b)))
    def poly_scalar_mul(self, m, a, s): return self._h(self.e.poly_scalar_mul(m, self._d(a), int(s)))
    def poly_rns_scalar_mul(self, m, a, s): return self._h(self.e.poly_scalar_mul(m, self._d(a), [int(x) for x in s]))
    def poly_ntt(self, m, a): return self._h(self.e.ntt_(m, self._d(a)))
    def poly_intt(self, m, a): return self._h(self.e.intt_(m, self._d(a)))
    def poly_involution(self, a): return self._h(self.e.poly_involution(self._d(a)))
    def poly_cycle(self, a, step): return self._h(self.e.poly_cycle(self._d(a), step))
    def mult_low_level(self, q, c1, c2): return self._h(self.e.mult_low_level(q, self._d(c1), self._d(c2)))
    def ext_prod(self, mext, pt, key): return self._h(self.e.ext_prod(mext, self._d(pt), self.e.to_device(key)))
    def ckks_rescale(self, m, ct): return self._h(self.e.ckks_rescale(m, self._d(ct)))
    def bgv_mod_drop(self, m, t, ct): return self._h(self.e.bgv_mod_switch(m, t, self._d(ct)))
    def ckks_relinearize(self, mext, quad, key): return self._h(self.e.ckks_relinearize(mext, self._d(quad), self.e.to_device(key)))
    def bgv_relinearize(self, mext, quad, key): return self._h(self.e.bgv_relinearize(mext, self._d(quad), self.e.to_device(key)))
    def ckks_rotate(self, mext, ct, key, step): return self._h(self.e.ckks_rotate(mext, self._d(ct), self.e.to_device(key), step))
    def ckks_conjugate(self, mext, ct, key): return self._h(self.e.ckks_conjugate(mext, self._d(ct), self.e.to_device(key)))
    def ckks_mult(self, mext, c1, c2, key): return self._h(self.e.ckks_mult(mext, self._d(c1), self._d(c2), self.e.to_device(key)))
    def bgv_mult(self, mext, t, c1, c2, key): return self._h(self.e.bgv_mult(mext, t, self._d(c1), self._d(c2), self.e.to_device(key)))

    def rlwe_encrypt_core(self, m, noise, c1, pt, sk):
        import torch

        d_noise = torch.from_numpy(np.ascontiguousarray(noise)[None]).to(f"cuda:{self.e.device}")
        return self._h(self.e.rlwe_encrypt_core(m, d_noise, self._d(c1), self._d(pt), self.e.to_device(sk)))

    def rlwe_decrypt_core(self, m, ct, sk): return self._h(self.e.rlwe_decrypt_core(m, self._d(ct), self.e.to_device(sk)))
    def rns_base_from_single(self, old, new, x): return self._h(self.e.rns_base_from_single(old, new, self._d(x)))

    def rns_base_to_single_small(self, m, t, x):
        out, flags = self.e.rns_base_to_single_small(m, t, self._d(x))
        return not bool(flags.cpu()[0]), self._h(out)

    def rns_base_to_single(self, m, t, x): return self._h(self.e.rns_base_to_single(m, t, self._d(x)))



def main():
    from cases import run_cases
    from hehub_amd.engine import Engine

    eng = Engine(0)
    try:
        lib = EngineLib(eng)
        res = run_cases(lib, big=True, fnv=lib.fnv)
    finally:
        eng.close()
    with open("/proc/self/maps") as f:
        maps = f.read()
    loaded = {name: (name in maps) for name in ("libhehub_amd.so", "libhehub_oracle", "libhehub_ref")}
    json.dump({"cases": res, "loaded": loaded, "oracle_module_imported": "oracle.pyoracle" in sys.modules}, sys.stdout)


if __name__ == "__main__":
    main()
