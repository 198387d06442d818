"""Randomised bit-exact parity sweep on a GPU: random ring sizes, moduli sets, batch sizes and entry points, every
result compared word for word with the oracle (test infrastructure).    gpurun -- 'python tools/fuzz_parity.py 300'
Argument: seconds to run (default 120).  Exits non-zero on the first mismatch and prints the case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")   # numpy <-> device copies through staging buffers (tests/conftest.py)
import params as P  # noqa: E402
from hehub_amd.engine import Engine  # noqa: E402
from hehub_amd.sharded import ShardedMult  # noqa: E402
from oracle.pyoracle import Oracle, SplitMix  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
HKS = len(sys.argv) > 3 and sys.argv[3] == "hks"   # third argument "hks": only the hybrid key switch against its exact integer model
LEVELA = len(sys.argv) > 3 and sys.argv[3] == "levela"   # "levela": the pipelines at parity level A (expected = the oracle's words mod q
                                                         # where the ring degree has tiled kernels, the raw words elsewhere) + residue transforms
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc, eng = Oracle("orc"), Engine(0)
if LEVELA:
    eng.set_parity_level("A")
pool = P.P40 + P.P50          # every list prime supports 2N | q-1 up to N = 32768
t0, cases, rs = time.time(), 0, np.random.RandomState(seed0)
hist = {}


def canon(moduli, a):
    """level A: canonical residue of every word of a [..., L', n] array whose limbs are the first L' of `moduli`"""
    return a % np.array(moduli[:a.shape[-2]], dtype=np.uint64)[:, None]


def check(name, got, exp, **kw):
    if LEVELA and name not in ("ntt", "intt") and 11 <= kw.get("logn", 0) <= 15:
        exp = canon(kw["mext"], exp)
    if not np.array_equal(got, exp):
        print("MISMATCH", name, kw, "first bad index", np.argwhere(got != exp)[:3].tolist())
        sys.exit(1)


while time.time() - t0 < budget:
    logn = int(rs.choice([1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 15], p=[.04, .04, .05, .07, .1, .1, .15, .15, .12, .1, .08]))
    n = 1 << logn
    L = int(rs.randint(1, 7)) if logn < 14 else int(rs.randint(1, 4))
    B = int(rs.randint(1, 12)) if logn < 13 else int(rs.randint(1, 4))   # up to 11: ragged last groups of the multi-limb inverse workgroups
    idx = rs.choice(len(pool), L + 1, replace=False)
    mext = [pool[i] for i in idx]
    q = mext[:L]
    rng = SplitMix(int(rs.randint(1, 1 << 30)))
    kw = dict(logn=logn, L=L, B=B, mext=mext)
    op = rs.choice(["ntt", "elem", "perm", "base", "mult", "bgv", "rot", "drop", "encdec", "sharded", "hks"], p=[.1] * 10 + [0.0]) if not HKS else "hks"
    if LEVELA:
        op = rs.choice(["resid", "mult", "bgv", "rot", "drop"], p=[.2, .25, .2, .2, .15])
    if op == "resid":
        if not 11 <= logn <= 15:
            continue
        x = np.stack([rng.poly((L, n), [2 * m for m in q]) for _ in range(B)])        # lazy input words
        d = eng.to_device(x); eng.ntt_residues_(q, d)
        y = np.stack([orc.poly_ntt(q, x[i]) for i in range(B)])
        check("ntt", eng.to_host(d), canon(q, y), **kw)
        d = eng.to_device(y); eng.intt_residues_(q, d)                                # the reference's lazy forward words back
        check("intt", eng.to_host(d), np.stack([orc.poly_reduce_strict(q, orc.poly_intt(q, y[i])) for i in range(B)]), **kw)
    elif op == "ntt":
        x = np.stack([rng.poly((L, n), q) for _ in range(B)])
        d = eng.to_device(x); eng.ntt_(q, d)
        y = np.stack([orc.poly_ntt(q, x[i]) for i in range(B)])
        check("ntt", eng.to_host(d), y, **kw)
        strict = bool(rs.randint(2))
        eng.intt_(q, d, strict=strict)
        z = np.stack([orc.poly_intt(q, y[i]) for i in range(B)])
        if strict:
            z = np.stack([orc.poly_reduce_strict(q, z[i]) for i in range(B)])
        check("intt", eng.to_host(d), z, strict=strict, **kw)
    elif op == "elem":
        a = np.stack([rng.poly((L, n), [2 * m for m in q]) for _ in range(B)])   # lazy operands in [0, 2q)
        b = np.stack([rng.poly((L, n), [2 * m for m in q]) for _ in range(B)])
        da, db = eng.to_device(a), eng.to_device(b)
        check("add", eng.to_host(eng.poly_add(q, da, db)), np.stack([orc.poly_add(q, a[i], b[i]) for i in range(B)]), **kw)
        check("sub", eng.to_host(eng.poly_sub(q, da, db)), np.stack([orc.poly_sub(q, a[i], b[i]) for i in range(B)]), **kw)
        check("mul", eng.to_host(eng.poly_mul(q, da, db)), np.stack([orc.poly_mul(q, a[i], b[i]) for i in range(B)]), **kw)
        sc = [int(w) for w in rng.words(L)]
        check("smul", eng.to_host(eng.poly_scalar_mul(q, da, sc)), np.stack([orc.poly_rns_scalar_mul(q, a[i], sc) for i in range(B)]), **kw)
    elif op == "base":
        t = int(rs.choice([2, 257, 65537, 786433, P.P50[3]]))
        x1 = np.stack([rng.words(n, 2 * t) for _ in range(B)])                       # one modulus -> many (lazy input)
        check("from_single", eng.to_host(eng.rns_base_from_single(t, q, eng.to_device(x1))),
              np.stack([orc.rns_base_from_single(t, q, x1[i]) for i in range(B)]), t=t, **kw)
        if t % 2 and t not in q:
            xs = np.stack([P.small_rns_poly(rng, n, q) if rs.randint(2) else rng.poly((L, n), q) for _ in range(B)])
            check("to_single", eng.to_host(eng.rns_base_to_single(q, t, eng.to_device(xs))),
                  np.stack([orc.rns_base_to_single(q, t, xs[i]) for i in range(B)]), t=t, **kw)
    elif op == "hks":
        from test_hks import model_switch
        logn = int(rs.choice([1, 2, 3, 4, 5, 11], p=[.1, .15, .2, .25, .25, .05])); n = 1 << logn
        L = int(rs.randint(1, 7)); alpha = int(rs.randint(1, min(L, 8) + 1)); k = int(rs.randint(1, 5))
        idx = rs.choice(len(pool), L + k, replace=False)
        mext = [pool[i] for i in idx]
        nd = (L + alpha - 1) // alpha
        pt = rng.poly((L, n), mext[:L]); key = rng.poly((nd, 2, L + k, n), mext)
        got = eng.to_host(eng.hks_switch(mext, k, alpha, eng.to_device(pt[None]), eng.to_device(key)))[0]
        check("hks", got, model_switch(orc, logn, mext, L, k, alpha, pt, key), logn=logn, L=L, k=k, alpha=alpha, mext=mext)
    elif op == "perm":
        a = np.stack([rng.poly((L, n), q) for _ in range(B)])
        da = eng.to_device(a)
        step = int(rs.randint(0, max(1, n // 2)))
        check("cycle", eng.to_host(eng.poly_cycle(da, step)), np.stack([orc.poly_cycle(a[i], step) for i in range(B)]), step=step, **kw)
        check("invol", eng.to_host(eng.poly_involution(da)), np.stack([orc.poly_involution(a[i]) for i in range(B)]), **kw)
    else:
        if L < 2:
            continue
        ct1 = np.stack([rng.poly((2, L, n), q) for _ in range(B)]); ct2 = np.stack([rng.poly((2, L, n), q) for _ in range(B)])
        key = rng.poly((L, 2, L + 1, n), mext)
        d1, d2, dk = eng.to_device(ct1), eng.to_device(ct2), eng.to_device(key)
        if op == "mult":
            check("ckks_mult", eng.to_host(eng.ckks_mult(mext, d1, d2, dk)), np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) for i in range(B)]), **kw)
            pairs = [(d1[i, 0], d1[i, 1], d2[B - 1 - i, 0], d2[B - 1 - i, 1]) for i in range(B)]    # the fused pipeline, operands by address
            check("ckks_mult_rows", eng.to_host(eng.ckks_mult_rows(mext, L, pairs, dk)),
                  np.stack([orc.ckks_mult(mext, ct1[i], ct2[B - 1 - i], key) for i in range(B)]), **kw)
        elif op == "bgv":
            t = int(rs.choice([2, 257, 65537, 786433]))
            check("bgv_mult", eng.to_host(eng.bgv_mult(mext, t, d1, d2, dk)), np.stack([orc.bgv_mult(mext, t, ct1[i], ct2[i], key) for i in range(B)]), t=t, **kw)
            pairs = [(d1[i, 0], d1[i, 1], d2[B - 1 - i, 0], d2[B - 1 - i, 1]) for i in range(B)]
            check("bgv_mult_rows", eng.to_host(eng.bgv_mult_rows(mext, t, pairs, dk)),
                  np.stack([orc.bgv_mult(mext, t, ct1[i], ct2[B - 1 - i], key) for i in range(B)]), t=t, **kw)
        elif op == "rot":
            step = int(rs.randint(0, max(1, n // 2)))
            check("rotate", eng.to_host(eng.ckks_rotate(mext, d1, dk, step)), np.stack([orc.ckks_rotate(mext, ct1[i], key, step) for i in range(B)]), step=step, **kw)
            check("conj", eng.to_host(eng.ckks_conjugate(mext, d2, dk)), np.stack([orc.ckks_conjugate(mext, ct2[i], key) for i in range(B)]), **kw)
            # a key, a step and the operand addresses PER ciphertext (hp_dev_ckks_rotate_many / _rows): against the single calls
            keys2 = [key, rng.poly((L, 2, L + 1, n), mext)]
            dks = [dk, eng.to_device(keys2[1])]
            which = [int(rs.randint(2)) for _ in range(B)]
            steps = [int(rs.randint(0, max(1, n // 2))) for _ in range(B)]
            cj = [bool(rs.randint(4) == 0) for _ in range(B)]
            exp = np.stack([orc.ckks_conjugate(mext, ct1[i], keys2[which[i]]) if cj[i] else orc.ckks_rotate(mext, ct1[i], keys2[which[i]], steps[i])
                            for i in range(B)])
            check("rot_many", eng.to_host(eng.ckks_rotate_many(mext, L, d1, [dks[w] for w in which], steps, cj)), exp, steps=steps, **kw)
            src = [int(rs.randint(B)) for _ in range(B)]           # ciphertexts picked (with repetition) by address
            polys = [(d1[i, 0], d1[i, 1]) for i in src]
            exp = np.stack([orc.ckks_conjugate(mext, ct1[src[i]], keys2[which[i]]) if cj[i] else orc.ckks_rotate(mext, ct1[src[i]], keys2[which[i]], steps[i])
                            for i in range(B)])
            check("rot_rows", eng.to_host(eng.ckks_rotate_many_rows(mext, L, polys, [dks[w] for w in which], steps, cj)), exp, steps=steps, **kw)
        elif op == "drop":
            t = int(rs.choice([2, 257, 65537]))
            check("rescale", eng.to_host(eng.ckks_rescale(q, d1)), np.stack([orc.ckks_rescale(q, ct1[i]) for i in range(B)]), **kw)
            check("modsw", eng.to_host(eng.bgv_mod_switch(q, t, d2)), np.stack([orc.bgv_mod_drop(q, t, ct2[i]) for i in range(B)]), t=t, **kw)
        elif op == "encdec":
            import torch
            cs = [P.edge_case(rng, logn, q) for _ in range(B)]
            sk = cs[0][3]
            noise = np.stack([c[0] for c in cs]); c1 = np.stack([c[1] for c in cs]); pt = np.stack([c[2] for c in cs])
            ct = eng.rlwe_encrypt_core(q, torch.from_numpy(noise).to("cuda:0"), eng.to_device(c1), eng.to_device(pt), eng.to_device(sk))
            exp = np.stack([orc.rlwe_encrypt_core(q, noise[i], c1[i], pt[i], sk) for i in range(B)])
            check("enc", eng.to_host(ct), exp, **kw)
            check("dec", eng.to_host(eng.rlwe_decrypt_core(q, ct, eng.to_device(sk))), np.stack([orc.rlwe_decrypt_core(q, exp[i], sk) for i in range(B)]), **kw)
        else:
            world = int(rs.randint(1, 6))
            t = int(rs.choice([0, 65537]))
            sm = ShardedMult(eng, mext, world, plain_modulus=t)
            bufs = sm.buffers(B, n)
            for b in bufs.values():
                b.fill_(-1)
            gens = [sm.stages(r, d1, d2, dk, bufs) for r in range(world)]
            while any([next(g, None) is not None for g in gens]):
                pass
            exp = np.stack([orc.ckks_mult(mext, ct1[i], ct2[i], key) if t == 0 else orc.bgv_mult(mext, t, ct1[i], ct2[i], key) for i in range(B)])
            check("sharded", eng.to_host(bufs["out"]), exp, world=world, t=t, **kw)
    cases += 1
    hist[(op, logn)] = hist.get((op, logn), 0) + 1
print(f"fuzz ok: {cases} random cases in {time.time() - t0:.0f} s (seed {seed0})")
ops = sorted({k[0] for k in hist})
for o in ops:
    print(f"  {o:8s}", " ".join(f"2^{l}:{hist[(o, l)]}" for l in sorted({k[1] for k in hist if k[0] == o})))
