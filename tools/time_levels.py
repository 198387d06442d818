"""C3 / C5 hom-mult step time at parity level B and A on the same buffers: python tools/time_levels.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import params as P
from hehub_amd.engine import Engine
from test_gpu_full_batch import rand_dev

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = Engine(0)
for name, logn, mext, B, bgv in (("C3 ckks", P.C3_LOGN, P.C3_MODULI_EXT, 256, False), ("C5 bgv", P.C5_LOGN, P.C5_MODULI_EXT, 512, True)):
    n, L = 1 << logn, len(mext) - 1
    ct1 = rand_dev(eng, (B, 2, L, n), mext[:L], 1); ct2 = rand_dev(eng, (B, 2, L, n), mext[:L], 2); key = rand_dev(eng, (L, 2, L + 1, n), mext, 3)
    out = eng.empty((B, 2, L - 1, n))
    f = (lambda: eng.bgv_mult(mext, P.C5_T, ct1, ct2, key, out=out)) if bgv else (lambda: eng.ckks_mult(mext, ct1, ct2, key, out=out))
    for rep in range(2):
        for lvl in ("B", "A"):
            eng.set_parity_level(lvl)
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
            eng.prof_begin("ntt,intt,ntt_drop,tensor,ks_inner"); f(); torch.cuda.synchronize(); nl, ms = eng.prof_end()
            print(f"{name} level {lvl}: {1e3 * dt:.3f} ms per step, {B / dt:,.0f} hom-mult/s   (kernel families {ms:.3f} ms in {nl} launches)", flush=True)
eng.close()
