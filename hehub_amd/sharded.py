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
    """The two exchanges of the mode over torch.distributed."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        if not dist.is_initialized():       # single process: both exchanges are the identity
            self.world, self.rank, self.staged = 1, 0, False
            return
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"   # gloo has no device all_gather: go through host memory

    def all_gather_limbs(self, buf, ranges):
        """buf: [rows][limbs][n]; rank r holds valid data in buf[:, ranges[r][0]:ranges[r][1]] and ends up with all."""
        import torch

        kmax = max(hi - lo for lo, hi in ranges)
        if kmax == 0 or self.world == 1:
            return
        rows, _, n = buf.shape
        lo, hi = ranges[self.rank]
        send = torch.zeros((rows, kmax, n), dtype=buf.dtype, device=buf.device)
        send[:, : hi - lo] = buf[:, lo:hi]
        if self.staged:
            send_h = send.cpu()
            recv_h = [torch.empty_like(send_h) for _ in range(self.world)]
            self.dist.all_gather(recv_h, send_h, group=self.group)
            recv = [t.to(buf.device) for t in recv_h]
        else:
            flat = torch.empty((self.world,) + tuple(send.shape), dtype=buf.dtype, device=buf.device)
            self.dist.all_gather_into_tensor(flat, send, group=self.group)
            recv = [flat[r] for r in range(self.world)]
        for r, (a, b) in enumerate(ranges):
            if r != self.rank and b > a:
                buf[:, a:b] = recv[r][:, : b - a]

    def broadcast(self, t, src: int):
        if self.world == 1:
            return
        if self.staged:
            h = t.cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            self.dist.broadcast(t, src=src, group=self.group)


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

        e._chk(lib.hp_dev_ks_inner_range(h, logn, L, m_ext, B, k0, k1, P(coef), d2, 3 * L, P(key), P(ks)))
        own_p = owner_of(L, self.ranges)
        if rank == own_p:
            e._chk(lib.hp_dev_drop_coeffs(h, logn, L + 1, m_ext, inner_t, 2 * B, P(ks), P(c_p)))
        yield ("broadcast", c_p, own_p)

        e._chk(lib.hp_dev_drop_apply_range(h, logn, L + 1, m_ext, inner_t, 2 * B, a0, a1, P(ks), P(c_p), P(quad), L, 3 * L, 3,
                                           P(relin)))
        own_q = owner_of(L - 1, self.ranges)
        if rank == own_q:
            e._chk(lib.hp_dev_drop_coeffs(h, logn, L, m_ct, t, 2 * B, P(relin), P(c_q)))
        yield ("broadcast", c_q, own_q)

        e._chk(lib.hp_dev_drop_apply_range(h, logn, L, m_ct, t, 2 * B, b0, b1, P(relin), P(c_q), None, 0, 0, 0, P(out)))
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
