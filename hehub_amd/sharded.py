"""Limb-sharded ("latency") mode: ONE ciphertext multiplication cut across the GPUs of a node.

SURVEY.md section 8e / BASELINE.json north_star: "RCCL all-gather over xGMI only for the key-switch
accumulation".  The batch mode (bench.py, hehub_amd/dist.py) needs no collective at all; this module is the
other way to use N GPUs: every rank works on the SAME ciphertexts and owns a contiguous range of the
extended moduli q_0..q_{L-1}, p.  All arithmetic for output modulus k happens on the owner of k, in the
reference's order, so the result is bit-identical to the single-GPU path (parity Level B).

Per multiplication (ckks::mult_low_level + relinearize + rescale_inplace, or the BGV composition):

  stage 1  tensor product on the owned limbs; c[j] = strict(INTT(d2[j])) for the owned digits j   (local)
  exchange ALL-GATHER of the coefficient-form digit limbs c[0..L)          L*8N bytes per ciphertext
  stage 2  digits D[j][k] = NTT_k(c[j]) and the u128 inner product for the owned k                (local)
           owner of p:  c_p = strict(INTT_p(ks[.][p]))
  exchange BROADCAST of c_p from the owner of p                             2*8N bytes per ciphertext
  stage 3  drop p on the owned limbs, += d0, d1;  owner of q_{L-1}: c_q = strict(INTT(relin[.][L-1]))
  exchange BROADCAST of c_q from the owner of q_{L-1}                       2*8N bytes per ciphertext
  stage 4  drop q_{L-1} on the owned limbs
  (optional) ALL-GATHER of the result limbs so that every rank holds the whole ciphertext

torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPUs; "gloo" is staged through
host memory and exists for tests).  The stages themselves are C-ABI calls (include/hehub_amd.h, "limb-range
stages"); nothing is computed by torch.

Ownership: contiguous ranges of the L+1 extended moduli, sizes differing by at most one, larger ranges first -- the
special prime (L digit transforms per polynomial instead of L-1, the most expensive output modulus) is the last
modulus and so always sits in a smallest range.  The speed-up over one GPU is bounded by (L+1) / ceil((L+1)/W):
11 moduli over 8 GPUs = 2,2,2,1,1,1,1,1 -> at most 5.5 x (stated in DESIGN.md section 6; the throughput mode is the
batch-sharded one).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

from .engine import Engine, _u64arr


def limb_ranges(n_limbs: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ownership ranges of `n_limbs` indices over `world` ranks, sizes differing by at most one."""
    base, extra = divmod(n_limbs, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def owner_of(k: int, ranges: Sequence[Tuple[int, int]]) -> int:
    for r, (lo, hi) in enumerate(ranges):
        if lo <= k < hi:
            return r
    raise ValueError(k)


def clip(ranges, limit):
    return [(min(lo, limit), min(hi, limit)) for lo, hi in ranges]


class Comm:
    """The two exchanges of the mode over torch.distributed, one process per GPU.

    All-gather of owned limbs = every rank SENDS its slice to every peer and RECEIVES theirs (batched isend / irecv: with
    the "nccl" backend = RCCL each transfer goes straight over the xGMI link between the two GPUs -- no ring, so a shard
    crosses exactly one link), exact sizes (no padding to the largest slice), through staging tensors that are allocated
    once per buffer shape and kept.  "gloo" (tests) stages through host memory.  (In ONE process the same exchange is
    done with direct peer writes by the C node layer, hehub_amd/csrc/hp_node.cpp.)"""

    TRANSPORTS = ("p2p", "allgather")

    def __init__(self, group=None, transport=None):
        """transport: "p2p" (default) = the batched isend / irecv exchange described above; "allgather" = ONE collective
        (torch.distributed.all_gather = ncclAllGather on RCCL), the form BASELINE's north star names -- every rank contributes a
        slice padded to the largest one (a collective needs equal counts: 11 moduli over 8 ranks send 2 limbs from every rank,
        16 instead of 11 limbs on the wire) and RCCL runs its ring / tree over the links.  HP_SHARDED_TRANSPORT selects it from
        the environment, `bench.py --workload ckks-limb --limb-transport allgather` from the command line: the day an 8-GPU node
        is at hand, p2p against the collective is one A/B run (DESIGN.md section 6 says why p2p is the default)."""
        import os

        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.transport = transport or os.environ.get("HP_SHARDED_TRANSPORT", "p2p")
        if self.transport not in self.TRANSPORTS:
            raise ValueError(f"unknown transport {self.transport!r}: one of {self.TRANSPORTS}")
        self._stage = {}
        self.force = bool(os.environ.get("HP_SHARDED_FORCE_COLLECTIVES")) and dist.is_initialized()
        if not dist.is_initialized():       # single process: both exchanges are the identity
            self.world, self.rank, self.staged = 1, 0, False
            return
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"   # gloo has no device transfers: go through host memory

    def _buffers(self, buf, ranges):
        """persistent contiguous staging per (buffer, ownership): [rows][owned limbs][n] to send, one per peer to receive"""
        import torch

        key = (buf.data_ptr(), tuple(buf.shape), tuple(ranges), self.staged)
        st = self._stage.get(key)
        if st is None:
            rows, _, n = buf.shape
            dev = "cpu" if self.staged else buf.device
            st = {r: torch.empty((rows, hi - lo, n), dtype=buf.dtype, device=dev) for r, (lo, hi) in enumerate(ranges) if hi > lo}
            if len(self._stage) > 16:
                self._stage.clear()
            self._stage[key] = st
        return st

    def _gather_collective(self, buf, ranges):
        """the same exchange as one all_gather of equal-sized (padded) slices"""
        import torch

        rows, _, n = buf.shape
        wmax = max(hi - lo for lo, hi in ranges)
        key = ("ag", buf.data_ptr(), tuple(buf.shape), tuple(ranges), self.staged)
        st = self._stage.get(key)
        if st is None:
            dev = "cpu" if self.staged else buf.device
            st = (torch.zeros((rows, wmax, n), dtype=buf.dtype, device=dev),
                  [torch.empty((rows, wmax, n), dtype=buf.dtype, device=dev) for _ in range(self.world)])
            if len(self._stage) > 16:
                self._stage.clear()
            self._stage[key] = st
        send, recv = st
        lo, hi = ranges[self.rank]
        if hi > lo:
            send[:, :hi - lo].copy_(buf[:, lo:hi])
        self.dist.all_gather(recv, send, group=self.group)
        for r, (a, b) in enumerate(ranges):
            if r != self.rank and b > a:
                buf[:, a:b].copy_(recv[r][:, :b - a])

    def all_gather_limbs(self, buf, ranges):
        """buf: [rows][limbs][n]; rank r holds valid data in buf[:, ranges[r][0]:ranges[r][1]] and ends up with all."""
        # (one rank: the identity -- unless HP_SHARDED_FORCE_COLLECTIVES asks for the calls anyway, tests/test_gpu_rccl.py)
        if (self.world == 1 and not self.force) or max(hi - lo for lo, hi in ranges) == 0:
            return
        if self.transport == "allgather":
            return self._gather_collective(buf, ranges)
        dist = self.dist
        st = self._buffers(buf, ranges)
        lo, hi = ranges[self.rank]
        ops = []
        if hi > lo:
            st[self.rank].copy_(buf[:, lo:hi])                      # contiguous copy of the owned slice
            ops += [dist.P2POp(dist.isend, st[self.rank], self._peer(d), self.group) for d in range(self.world) if d != self.rank]
        ops += [dist.P2POp(dist.irecv, st[r], self._peer(r), self.group) for r in st if r != self.rank]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for r, (a, b) in enumerate(ranges):
            if r != self.rank and b > a:
                buf[:, a:b].copy_(st[r])

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def broadcast(self, t, src: int):
        if self.world == 1 and not self.force:
            return
        if self.staged:
            h = t.cpu()
            self.dist.broadcast(h, src=self._peer(src), group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            self.dist.broadcast(t, src=self._peer(src), group=self.group)


class ShardedMult:
    """ckks::mult + relinearize + rescale_inplace (plain_modulus = 0) or bgv::mult_low_level + relinearize +
    mod_switch_inplace (plain_modulus = t) on ciphertext batches replicated on every rank, cut by output modulus."""

    def __init__(self, eng: Engine, moduli_ext: Sequence[int], world: int, plain_modulus: int = 0):
        self.eng = eng
        self.mext = [int(q) for q in moduli_ext]
        self.L = len(self.mext) - 1
        self.world = world
        self.t = int(plain_modulus)
        self.ranges = limb_ranges(self.L + 1, world)

    def buffers(self, B: int, n: int):
        e, L = self.eng, self.L
        return {"quad": e.empty((B, 3, L, n)), "coef": e.empty((B, L, n)), "ks": e.empty((B, 2, L + 1, n)),
                "c_p": e.empty((2 * B, n)), "relin": e.empty((B, 2, L, n)), "c_q": e.empty((2 * B, n)),
                "out": e.empty((B, 2, L - 1, n))}

    def stages(self, rank: int, ct1, ct2, key, bufs, gather_out: bool = True):
        """Generator: runs the local stages of `rank` and yields the exchange that must happen before it resumes:
        ("all_gather", tensor[rows][limbs][n], ranges) or ("broadcast", tensor, src)."""
        e, lib, h, L, mext, t = self.eng, self.eng.lib, self.eng.h, self.L, self.mext, self.t
        B, _, _, n = ct1.shape
        logn = n.bit_length() - 1
        m_ext, m_ct = _u64arr(mext), _u64arr(mext[:L])
        k0, k1 = self.ranges[rank]
        a0, a1 = min(k0, L), min(k1, L)            # owned ciphertext limbs
        b0, b1 = min(k0, L - 1), min(k1, L - 1)    # ... that survive the final drop
        inner_t = 1 if t else 0                    # bgv.h:32: relinearize's internal mod switch sees plain_modulus 1
        P = e._ptr
        quad, coef, ks, c_p, relin, c_q, out = (bufs[k] for k in ("quad", "coef", "ks", "c_p", "relin", "c_q", "out"))
        d2 = C.c_void_p(quad.data_ptr() + 2 * L * n * 8)   # polynomial 2 of the first quadratic ciphertext; stride 3L limbs

        e._chk(lib.hp_dev_mult_low_level_range(h, logn, L, m_ct, B, a0, a1, P(ct1), P(ct2), P(quad)))
        e._chk(lib.hp_dev_ks_coef_range(h, logn, L, m_ext, B, a0, a1, d2, 3 * L, P(coef)))
        yield ("all_gather", coef, clip(self.ranges, L))

        e._chk(lib.hp_dev_ks_inner_range_strict(h, logn, L, m_ext, B, k0, k1, P(coef), d2, 3 * L, P(key), P(ks)))
        own_p = owner_of(L, self.ranges)
        if rank == own_p:
            e._chk(lib.hp_dev_drop_coeffs(h, logn, L + 1, m_ext, inner_t, 2 * B, P(ks), P(c_p)))
        yield ("broadcast", c_p, own_p)

        e._chk(lib.hp_dev_drop_apply_range_strict(h, logn, L + 1, m_ext, inner_t, 2 * B, a0, a1, P(ks), P(c_p), P(quad), L, 3 * L, 3,
                                           P(relin)))
        own_q = owner_of(L - 1, self.ranges)
        if rank == own_q:
            e._chk(lib.hp_dev_drop_coeffs(h, logn, L, m_ct, t, 2 * B, P(relin), P(c_q)))
        yield ("broadcast", c_q, own_q)

        e._chk(lib.hp_dev_drop_apply_range_strict(h, logn, L, m_ct, t, 2 * B, b0, b1, P(relin), P(c_q), None, 0, 0, 0, P(out)))
        if gather_out:
            yield ("all_gather", out.view(2 * B, L - 1, n), clip(self.ranges, L - 1))

    def run(self, comm: Comm, ct1, ct2, key, bufs=None, gather_out: bool = True):
        """One multiplication on this rank's GPU; every rank must call it with the same ciphertexts."""
        assert comm.world == self.world
        B, _, _, n = ct1.shape
        bufs = bufs or self.buffers(B, n)
        for kind, tensor, arg in self.stages(comm.rank, ct1, ct2, key, bufs, gather_out):
            if kind == "all_gather":
                comm.all_gather_limbs(tensor, arg)
            else:
                comm.broadcast(tensor, arg)
        return bufs["out"]
