"""Host-side logic without a GPU: the table builders of hehub_amd/csrc/hp_tables.cpp.

* values: reference-order tables equal the oracle's (hence the reference's NTTFactors);
* layouts: a numpy model of the tiled kernels' dataflow (three passes over a 5-bit register index, the slot /
  class indexing documented in hp_ntt_fast.hip) driven by the kernel-order tables reproduces the oracle's
  transforms word for word -- so a layout mistake is caught here, before any GPU run."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import params as P
from oracle.pyoracle import SplitMix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "cpp", "libtables_shim.so")
U = np.uint64
M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def tb():
    src = [os.path.join(ROOT, "tests", "cpp", "tables_shim.cpp"), os.path.join(ROOT, "hehub_amd", "csrc", "hp_tables.cpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO] + src, check=True)
    lib = C.CDLL(SO)
    lib.tbl_root.restype = C.c_uint64
    lib.tbl_inverse.restype = C.c_uint64
    for f in ("tbl_root", "tbl_check"):
        getattr(lib, f).argtypes = [C.c_uint64, C.c_size_t]
    lib.tbl_inverse.argtypes = [C.c_uint64, C.c_uint64]
    lib.tbl_consts.argtypes = [C.c_uint64, C.c_void_p]
    for f in ("tbl_fwd_ref", "tbl_inv_ref", "tbl_fwd_fast", "tbl_inv_fast"):
        getattr(lib, f).argtypes = [C.c_uint64, C.c_size_t, C.c_void_p]
    return lib


def table(lib, name, q, logn, pairs):
    out = np.zeros((pairs, 2), dtype=U)
    getattr(lib, name)(q, logn, out.ctypes.data_as(C.c_void_p))
    return out


def test_reference_order_tables_and_constants(tb, orc):
    for q, logn in [(65537, 4), (P.C1_Q, 12), (P.P40[0], 11), (P.P50[0], 13)]:
        n = 1 << logn
        s, h = orc.ntt_factors(q, logn)
        t = table(tb, "tbl_fwd_ref", q, logn, n)
        assert (t[:, 0] == s).all() and (t[:, 1] == h).all()
        s, h = orc.intt_factors(q, logn)
        t = table(tb, "tbl_inv_ref", q, logn, 2 * n)
        assert (t[:, 0] == s).all() and (t[:, 1] == h).all()
        assert tb.tbl_root(q, logn) == orc.unity_root(q, n)
        c = np.zeros(9, dtype=U)
        tb.tbl_consts(q, c.ctypes.data_as(C.c_void_p))
        assert int(c[0]) == q and int(c[1]) == 2 * q and int(c[2]) == (-q) & M64
        assert int(c[3]) == orc._mqinv(q) and int(c[4]) == orc._r64(q) and int(c[5]) == orc._hq(int(c[4]) % q, q)
        assert int(c[6]) == M64 // q and int(c[7]) == int(np.log2(q) + 0.5) and int(c[8]) == int(q >= (1 << int(c[7])))
    assert tb.tbl_check(65537, 16) == 1 and tb.tbl_check(1234567890111111111, 4) == 1 and tb.tbl_check(P.C1_Q, 15) == 0
    for e, p in [(65537, P.P40[0]), (P.P50[0], P.P40[3]), (1, P.P40[0])]:
        assert tb.tbl_inverse(e, p) == orc.inverse_mod_prime(e, p)


def bfly(lo, hi, w, wh, q):
    """vectorised lazy Harvey butterfly on python ints held in object arrays"""
    qhat = (hi * wh) >> 64
    t = (hi * w - qhat * q) & M64
    return (lo + t) & M64, (lo + 2 * q - t) & M64


def run_pass(x, fwd, slots, tbl, ncls, cls, q):
    """x: object array [threads, 32]; the slot / register mapping of hp_ntt_fast.hip::pass_slots"""
    for s in slots:
        lg = (s + 1).bit_length() - 1
        b = 4 - lg if fwd else lg
        idx = s + 1 - (1 << lg)
        tw = tbl[s * ncls + cls]                       # [threads, 2]
        w, wh = tw[:, 0].astype(object), tw[:, 1].astype(object)
        for o in range(1 << (4 - lg)):
            r = ((idx << (b + 1)) | o) if fwd else ((o << (b + 1)) | idx)
            x[:, r], x[:, r | (1 << b)] = bfly(x[:, r], x[:, r | (1 << b)], w, wh, q)


@pytest.mark.parametrize("logn", [11, 12, 14])
def test_kernel_order_tables_drive_the_tiled_dataflow(tb, orc, logn):
    q = P.P40[0] if logn < 14 else P.P50[0]
    n, a = 1 << logn, logn - 10
    pb, T, nblk = 5 - a, 1 << (logn - 5), 1 << a
    k = int(np.log2(q) + 0.5)
    fix = int(q >= (1 << k))
    x0 = SplitMix(logn).words(n, 2 * q)
    fwd_ref = table(tb, "tbl_fwd_ref", q, logn, n)
    fk = table(tb, "tbl_fwd_fast", q, logn, 31 * nblk + 31 * T)
    t = np.arange(T)
    # ---- forward: layouts A (strided), B (blocked), C (contiguous) ----
    iA = np.array([[((r >> pb) << 10) | (tt << pb) | (r & ((1 << pb) - 1)) for r in range(32)] for tt in t])
    iB = np.array([[((tt >> 5) << 10) | (m << 5) | (tt & 31) for m in range(32)] for tt in t])
    iC = np.array([[(tt << 5) | r for r in range(32)] for tt in t])
    flat = x0.astype(object)
    xa = flat[iA]
    run_pass(xa, True, range((1 << a) - 1), fwd_ref[1:], 1, np.zeros(T, dtype=int), q)
    flat = np.empty(n, dtype=object); flat[iA] = xa
    xb = flat[iB]
    run_pass(xb, True, range(31), fk, nblk, t >> 5, q)
    flat = np.empty(n, dtype=object); flat[iB] = xb
    xc = flat[iC]
    run_pass(xc, True, range(31), fk[31 * nblk:], T, t, q)
    flat = np.empty(n, dtype=object); flat[iC] = xc
    out = np.array([(v - ((v >> k) - fix) * q) & M64 for v in flat], dtype=U)
    assert (out == orc.ntt(logn, q, x0)).all()
    # ---- inverse: C', B', A' with the scale table of the reference-order inverse table ----
    inv_ref = table(tb, "tbl_inv_ref", q, logn, 2 * n)
    ik = table(tb, "tbl_inv_fast", q, logn, 31 + 31 * 32 + 31 * T)
    flat = out.astype(object)
    xc = flat[iC]
    run_pass(xc, False, range(31), ik, 1, np.zeros(T, dtype=int), q)
    flat = np.empty(n, dtype=object); flat[iC] = xc
    xb = flat[iB]
    run_pass(xb, False, range(31), ik[31:], 32, t & 31, q)
    flat = np.empty(n, dtype=object); flat[iB] = xb
    xa = flat[iA]
    run_pass(xa, False, range((1 << pb) - 1, 31), ik[31 + 31 * 32:], T, t, q)
    flat = np.empty(n, dtype=object); flat[iA] = xa
    res = []
    for i, v in enumerate(flat):
        v = (v - ((v >> k) - fix) * q) & M64
        w, wh = int(inv_ref[n + i, 0]), int(inv_ref[n + i, 1])
        res.append((v * w - ((v * wh) >> 64) * q) & M64)
    assert (np.array(res, dtype=U) == orc.intt(logn, q, out)).all()
