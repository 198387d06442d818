import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle.pyoracle import Oracle, SplitMix
from hehub_amd.engine import Engine
o=Oracle("orc"); e=Engine(0)
for mods in ([1099510054913, 1073479681, 1072496641, 1099507695617],[1099510054913, 1099507695617],[1073479681, 1072496641]):
    n=8; L=len(mods); B=2
    rng=SplitMix(5)
    ct=rng.poly((B,2,L,n),mods)
    got=e.to_host(e.ckks_rescale(mods,e.to_device(ct)))
    exp=np.stack([o.ckks_rescale(mods,ct[i]) for i in range(B)])
    print(mods, (got==exp).all(), (got==exp).reshape(B,2,L-1,n).all(axis=3).astype(int).tolist())
    # emulate with verified pieces
    last=ct[:,:,L-1,:].reshape(B*2,1,n).copy()
    c=e.to_host(e.intt_([mods[-1]],e.to_device(last),strict=True)).reshape(B,2,n)
    cexp=np.stack([[o.batched_reduce_strict(mods[-1],o.intt(3,mods[-1],ct[b,h,L-1])) for h in range(2)] for b in range(B)])
    print("  clast ok", (c==cexp).all())
