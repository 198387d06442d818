import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle.pyoracle import Oracle, SplitMix
from hehub_amd.engine import Engine
o=Oracle("orc"); e=Engine(0)
moduli=[17179672577, 17179410433, 17176854529]; n=8; L=3; B=5
rng=SplitMix(3*31+5)
a=rng.poly((B,L,n),[2*m for m in moduli])
y=np.stack([o.poly_ntt(moduli,a[i]) for i in range(B)])
z=e.to_host(e.intt_(moduli,e.to_device(y)))
zs=e.to_host(e.intt_(moduli,e.to_device(y),strict=True))
ze=np.stack([o.poly_intt(moduli,y[i]) for i in range(B)])
zse=np.stack([o.poly_reduce_strict(moduli,ze[i]) for i in range(B)])
am=a % np.array(moduli,dtype=np.uint64)[None,:,None]
print("z==ze",(z==ze).all(),"zs==zse",(zs==zse).all(),"zs==a%q",(zs==am).all(),"zse==a%q",(zse==am).all())
print((zs==zse).reshape(B,-1).all(axis=1))
print(zs[0,0],zse[0,0],z[0,0])
