import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle.pyoracle import Oracle, SplitMix
from hehub_amd.engine import Engine
o=Oracle("orc"); e=Engine(0)
mods=[1073479681, 1072496641]; n=8; L=2; B=2
q=mods[0]; ql=mods[1]
rng=SplitMix(5)
ct=rng.poly((B,2,L,n),mods)
got=e.to_host(e.ckks_rescale(mods,e.to_device(ct)))
for b in range(B):
  for h in range(2):
    x=[int(v) for v in ct[b,h,0]]
    g=[int(v) for v in got[b,h,0]]
    rem_used=[(x[i]-g[i]*ql)%q for i in range(n)]
    remc=o.batched_reduce_strict(q,o.intt(3,q,np.array(rem_used,dtype=np.uint64)))
    c=o.batched_reduce_strict(ql,o.intt(3,ql,ct[b,h,1]))
    rem_exp=[(int(v)%q + (q - ql%q if int(v)>=ql//2 else 0))%q for v in c]
    print(b,h,"rem coeffs used:",[int(v) for v in remc][:4],"expected:",rem_exp[:4])
    # which c would give it?
    for bb in range(B):
      for hh in range(2):
        c2=o.batched_reduce_strict(ql,o.intt(3,ql,ct[bb,hh,1]))
        r2=[(int(v)%q + (q - ql%q if int(v)>=ql//2 else 0))%q for v in c2]
        if r2==[int(v) for v in remc]: print("   matches clast of",bb,hh)
for b in range(B):
  for h in range(2):
    c=o.batched_reduce_strict(ql,o.intt(3,ql,ct[b,h,1]))
    print("expected clast",b,h,[int(v) for v in c][:2])
