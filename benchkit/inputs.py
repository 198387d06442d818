"""Synthetic inputs: uniform residues generated on the device, and batches that repeat a few distinct classes so that
EVERY output of a timed buffer can be compared with a handful of CPU-checker evaluations."""
from __future__ import annotations


def rand_words(torch, shape, moduli, device, seed):
    """uniform words in [0, q_k) per limb (limb axis = -2), generated on the device"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(shape, dtype=torch.int64, device=device)
    for k, q in enumerate(moduli):
        out.select(-2, k).copy_(torch.randint(0, int(q), out.select(-2, k).shape, generator=g, device=device,
                                              dtype=torch.int64))
    return out


class Batch:
    """A batch of B items that repeats `period` distinct random classes (item i = class i mod period); period 0: all
    items distinct.  `base` keeps the classes (device), `full` is the batch the kernels see."""

    def __init__(self, torch, B, item_shape, moduli, device, seed, period):
        self.torch = torch
        self.period = period if 0 < period < B else 0
        if self.period:
            self.base = rand_words(torch, (self.period,) + tuple(item_shape), moduli, device, seed)
            idx = torch.arange(B, device=device) % self.period
            self.full = self.base.index_select(0, idx).contiguous()
        else:
            self.full = rand_words(torch, (B,) + tuple(item_shape), moduli, device, seed)
            self.base = None
        self.B = B

    def fresh(self):
        """a new copy of the periodic batch (for one more application of an in-place operation); periodic batches only"""
        torch = self.torch
        return self.base.index_select(0, torch.arange(self.B, device=self.base.device) % self.period).contiguous()

    def classes(self, sample=(0,)):
        """(indices of the items that stand for all others, their host copies as uint64)"""
        import numpy as np

        if self.period:
            return list(range(self.period)), self.base.cpu().numpy().view(np.uint64)
        idx = sorted({i % self.B for i in sample})
        return idx, self.full[idx].cpu().numpy().view(np.uint64)


def compare_classes(torch, out, expected_host, period, idx):
    """every item of `out` against its class (period > 0) or the sampled items against theirs; returns (ok, compared)"""
    import numpy as np

    exp = torch.from_numpy(np.ascontiguousarray(expected_host).view(np.int64)).to(out.device)
    if period:
        ok = all(bool(torch.equal(out[c::period], exp[c].expand_as(out[c::period]))) for c in range(period))
        return ok, out.shape[0]
    ok = all(bool(torch.equal(out[i], exp[j])) for j, i in enumerate(idx))
    return ok, len(idx)
