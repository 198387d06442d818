"""Wire / on-disk format of ring elements (include/hehub_amd.h "wire", hehub_amd/csrc/hp_wire.cpp).

Thin ctypes wrappers: the bytes are produced and validated by the library, not by Python."""
from __future__ import annotations

import ctypes as C
import struct
from typing import Sequence, Tuple

import numpy as np

from . import capi

POLY, CT, QUAD_CT, KSK = 1, 2, 3, 4


class WireDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("log_dimension", C.c_uint32), ("limbs", C.c_uint32), ("polys", C.c_uint32),
                ("rep_form", C.c_uint32), ("scheme_scalar", C.c_uint64)]


def _desc(kind, logn, limbs, polys, rep_form, scalar) -> WireDesc:
    if isinstance(scalar, float):
        scalar = struct.unpack("<Q", struct.pack("<d", scalar))[0]
    return WireDesc(kind, logn, limbs, polys, rep_form, int(scalar))


def pack(kind: int, moduli: Sequence[int], words: np.ndarray, rep_form: int = 1, scalar=0) -> bytes:
    """words: uint64 [polys][limbs][N] (a polynomial may omit the first axis)."""
    lib = capi.load()
    w = np.ascontiguousarray(words, dtype=np.uint64)
    if w.ndim == 2:
        w = w[None]
    polys, limbs, n = w.shape
    d = _desc(kind, n.bit_length() - 1, limbs, polys, rep_form, scalar)
    size = lib.hp_wire_bytes(C.byref(d))
    if size == 0:
        raise ValueError("invalid wire descriptor")
    buf = C.create_string_buffer(size)
    m = (capi.u64 * limbs)(*[int(q) for q in moduli])
    rc = lib.hp_wire_pack(C.byref(d), m, w.ctypes.data_as(capi.P), buf, size)
    if rc != capi.HP_OK:
        raise ValueError(f"hp_wire_pack rc={rc}")
    return buf.raw


def unpack(blob: bytes) -> Tuple[WireDesc, list, np.ndarray]:
    lib = capi.load()
    d = WireDesc()
    m = (capi.u64 * 32)()
    off = C.c_size_t(0)
    rc = lib.hp_wire_unpack(blob, len(blob), C.byref(d), m, 32, C.byref(off))
    if rc != capi.HP_OK:
        raise ValueError("not a valid HEHUBAMD object (magic, version, size or checksum)")
    n = 1 << d.log_dimension
    words = np.frombuffer(blob, dtype="<u8", count=d.polys * d.limbs * n, offset=off.value).reshape(d.polys, d.limbs, n)
    return d, [int(m[k]) for k in range(d.limbs)], words.astype(np.uint64)


def load_to_device(eng, blob: bytes):
    """Validated bytes -> device tensor [polys][limbs][N] (one host-to-device copy of the payload)."""
    d, moduli, _ = unpack(blob)
    t = eng.empty((d.polys, d.limbs, 1 << d.log_dimension))
    eng._chk(eng.lib.hp_dev_wire_load(eng.h, blob, len(blob), eng._ptr(t)))
    return d, moduli, t


def store_from_device(eng, kind: int, moduli: Sequence[int], t, rep_form: int = 1, scalar=0) -> bytes:
    polys, limbs, n = (1,) + tuple(t.shape) if t.dim() == 2 else tuple(t.shape)
    d = _desc(kind, n.bit_length() - 1, limbs, polys, rep_form, scalar)
    size = eng.lib.hp_wire_bytes(C.byref(d))
    buf = C.create_string_buffer(size)
    m = (capi.u64 * limbs)(*[int(q) for q in moduli])
    eng._chk(eng.lib.hp_dev_wire_store(eng.h, C.byref(d), m, eng._ptr(t), buf, size))
    return buf.raw
