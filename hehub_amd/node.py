"""ctypes front-end of the node layer (include/hehub_amd.h, "node"; hehub_amd/csrc/hp_node.cpp): several GPUs of one
node behind one handle, driven from ONE process through the C ABI -- the layer a C/C++ hehub application uses.
Host-resident numpy batches in, numpy out; nothing is computed here.  (The other way to use several GPUs, one process
per GPU over torch.distributed, is hehub_amd/dist.py and hehub_amd/sharded.py.)"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import capi
from .engine import HpError, InvalidArgument, _u64arr


class Node:
    def __init__(self, devices: Sequence[int]):
        import torch  # noqa: F401  (loads the HIP runtime the library binds to)

        if not torch.cuda.is_available():
            raise capi.EngineMissing("no HIP device visible: the node layer needs MI355X GPUs (no CPU fallback)")
        self.lib = capi.load()
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        h = capi.P()
        rc = self.lib.hp_node_create((C.c_int * self.world)(*self.devices), self.world, C.byref(h))
        if rc != capi.HP_OK:
            raise HpError(rc, "hp_node_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hp_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc == capi.HP_OK:
            return
        msg = self.lib.hp_node_last_error(self.h).decode()
        raise (InvalidArgument if rc == capi.HP_EINVAL else HpError)(rc, msg)

    def set_parity_level(self, level: str):
        """"B" (default) or "A" on every rank's context (include/hehub_amd.h: hp_ctx_set_parity_level)"""
        self._chk(self.lib.hp_node_set_parity_level(self.h, {"B": 0, "A": 1}[level.upper()]))

    TRANSPORTS = {"peer": 0, "rccl": 1, "packed": 2}   # include/hehub_amd.h: HP_TRANSPORT_*

    def set_transport(self, name: str):
        """how the limb-sharded plans made from now on exchange their limbs: "peer" (direct peer writes, default), "rccl" (ncclAllGather /
        ncclBroadcast through a communicator over the node's devices: every rank on its own device), "packed" (the collective's packed
        buffers moved by plain copies)"""
        self._chk(self.lib.hp_node_set_transport(self.h, self.TRANSPORTS[name]))

    def transport(self) -> str:
        return {v: k for k, v in self.TRANSPORTS.items()}[self.lib.hp_node_get_transport(self.h)]

    def peer_matrix(self) -> np.ndarray:
        """[a][b] = 1 when rank a writes rank b's device memory directly (hp_node_peer_matrix)"""
        m = (C.c_int * (self.world * self.world))()
        self._chk(self.lib.hp_node_peer_matrix(self.h, m))
        return np.array(m[:], dtype=np.int32).reshape(self.world, self.world)

    def pinned(self, shape) -> np.ndarray:
        """uint64 array in page-locked host memory (hp_host_alloc on rank 0's context); free with unpin()"""
        words = int(np.prod(shape))
        ptr = capi.P()
        ctx = self.lib.hp_node_ctx(self.h, 0)
        rc = self.lib.hp_host_alloc(ctx, words * 8, C.byref(ptr))
        if rc != capi.HP_OK:
            raise HpError(rc, "hp_host_alloc failed")
        buf = (C.c_uint64 * words).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = ptr
        return a

    def unpin(self, a: np.ndarray):
        ptr = self._pinned.pop(a.ctypes.data)
        self.lib.hp_host_free(self.lib.hp_node_ctx(self.h, 0), ptr)

    def slice(self, total: int, rank: int):
        lo, hi = capi.szt(0), capi.szt(0)
        self._chk(self.lib.hp_node_slice(self.h, total, rank, C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def replicate(self, words: np.ndarray):
        """host words -> one device copy per rank; returns the pointer array to pass as `key`"""
        words = np.ascontiguousarray(words)
        copies = (capi.P * self.world)()
        self._chk(self.lib.hp_node_replicate(self.h, words.ctypes.data_as(capi.P), words.size, copies))
        return copies

    def free_replicas(self, copies):
        self._chk(self.lib.hp_node_free_replicas(self.h, copies))

    def ckks_mult(self, moduli_ext, ct1: np.ndarray, ct2: np.ndarray, key, out: np.ndarray = None):
        B, _, L, n = ct1.shape
        out = np.empty((B, 2, L - 1, n), dtype=np.uint64) if out is None else out
        self._chk(self.lib.hp_node_ckks_mult_relin_rescale(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), B,
                                                           np.ascontiguousarray(ct1).ctypes.data_as(capi.P),
                                                           np.ascontiguousarray(ct2).ctypes.data_as(capi.P), key,
                                                           out.ctypes.data_as(capi.P)))
        return out

    def bgv_mult(self, moduli_ext, t: int, ct1: np.ndarray, ct2: np.ndarray, key):
        B, _, L, n = ct1.shape
        out = np.empty((B, 2, L - 1, n), dtype=np.uint64)
        self._chk(self.lib.hp_node_bgv_mult_relin_modswitch(self.h, n.bit_length() - 1, L, _u64arr(moduli_ext), t, B,
                                                            np.ascontiguousarray(ct1).ctypes.data_as(capi.P),
                                                            np.ascontiguousarray(ct2).ctypes.data_as(capi.P), key,
                                                            out.ctypes.data_as(capi.P)))
        return out

    def ntt_(self, moduli, x: np.ndarray, inverse: bool = False, strict: bool = False):
        B, L, n = x.shape
        assert x.flags["C_CONTIGUOUS"] and x.dtype == np.uint64
        self._chk(self.lib.hp_node_ntt(self.h, n.bit_length() - 1, L, _u64arr(moduli), B, x.ctypes.data_as(capi.P), int(inverse),
                                       int(strict)))
        return x


class ShardedPlan:
    """limb-sharded multiplication of one batch across the ranks of a node (hp_node_sharded_*)"""

    def __init__(self, node: Node, logn: int, moduli_ext, batch: int, plain_modulus: int = 0):
        self.node, self.logn, self.mext, self.batch = node, logn, [int(q) for q in moduli_ext], batch
        self.L = len(self.mext) - 1
        h = capi.P()
        node._chk(node.lib.hp_node_sharded_create(node.h, logn, self.L, _u64arr(self.mext), plain_modulus, batch, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.node.lib.hp_node_sharded_destroy(self.h)
            self.h = None

    def range(self, rank: int):
        k0, k1 = capi.szt(0), capi.szt(0)
        self.node._chk(self.node.lib.hp_node_sharded_range(self.h, rank, C.byref(k0), C.byref(k1)))
        return int(k0.value), int(k1.value)

    def mult(self, ct1: np.ndarray, ct2: np.ndarray, key):
        B, _, L, n = ct1.shape
        assert B == self.batch and L == self.L and n == 1 << self.logn
        out = np.empty((B, 2, L - 1, n), dtype=np.uint64)
        self.node._chk(self.node.lib.hp_node_sharded_mult(self.h, np.ascontiguousarray(ct1).ctypes.data_as(capi.P),
                                                          np.ascontiguousarray(ct2).ctypes.data_as(capi.P), key,
                                                          out.ctypes.data_as(capi.P)))
        return out

    def mult_dev(self, d_ct1, d_ct2, key, d_out):
        """operands already on the devices: per-rank lists of int64 torch tensors (device of the rank), result replicated into
        every rank's d_out tensor (hp_node_sharded_mult_dev)"""
        W = self.node.world
        arr = lambda ts: (capi.P * W)(*[C.c_void_p(t.data_ptr()) for t in ts])
        self.node._chk(self.node.lib.hp_node_sharded_mult_dev(self.h, arr(d_ct1), arr(d_ct2), key, arr(d_out)))
        return d_out


def dev_mult(node: Node, moduli_ext, d_ct1, d_ct2, key, d_out, plain_modulus: int = 0):
    """batch-sharded, device-resident: rank r multiplies the ciphertext pairs behind d_ct1[r] / d_ct2[r] (torch tensors
    [count_r][2][L][n] on the rank's device) into d_out[r] (hp_node_dev_ckks_mult_relin_rescale / _bgv_...)"""
    W = node.world
    counts = (capi.szt * W)(*[int(t.shape[0]) for t in d_ct1])
    L, n = int(d_ct1[0].shape[2]), int(d_ct1[0].shape[3])
    arr = lambda ts: (capi.P * W)(*[C.c_void_p(t.data_ptr()) for t in ts])
    if plain_modulus:
        rc = node.lib.hp_node_dev_bgv_mult_relin_modswitch(node.h, n.bit_length() - 1, L, _u64arr(moduli_ext), plain_modulus, counts,
                                                           arr(d_ct1), arr(d_ct2), key, arr(d_out))
    else:
        rc = node.lib.hp_node_dev_ckks_mult_relin_rescale(node.h, n.bit_length() - 1, L, _u64arr(moduli_ext), counts, arr(d_ct1),
                                                          arr(d_ct2), key, arr(d_out))
    node._chk(rc)
    return d_out
