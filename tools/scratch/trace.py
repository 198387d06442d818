import ctypes as C, numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ["HEHUB_AMD_LIB"]=os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e=Engine(0)
wl=sys.argv[1] if len(sys.argv)>1 else "15"
if wl=="15": logn,mods,B=15,P.C3_MODULI_EXT,256
else: logn,mods,B=14,P.C2_MODULI,1024
n=1<<logn; L=len(mods)
x=torch.randint(0,1<<40,(B,L,n),dtype=torch.int64,device="cuda")
for _ in range(3): e.ntt_(mods,x)
torch.cuda.synchronize()
lib=capi.load()
W=B*L; nrec=min(2048,(W+15)//16)*2
buf=np.zeros(nrec*12,dtype=np.uint64)
lib.hp_debug_trace.argtypes=[C.c_void_p,C.c_size_t]; lib.hp_debug_trace.restype=C.c_int
print("rc",lib.hp_debug_trace(buf.ctypes.data_as(C.c_void_p),buf.size))
t=buf.reshape(nrec,12).astype(np.int64)
t=t[t[:,0]>0]
d=np.diff(t[:,:10],axis=1)   # cycles per phase (100 MHz? s_memtime ticks)
names=["load","passA","exch1","passB","exch2","passC","fold","exch3","store"]
print("records",len(t))
for w,lab in ((0,"wave0"),(1,"lastwave")):
    dd=d[w::2]
    print(lab," ".join(f"{nm}={np.median(dd[:,i]):.0f}" for i,nm in enumerate(names)), "total=%.0f"%np.median(t[w::2,9]-t[w::2,0]))
print("span of starts (first..last WG):", t[:,0].max()-t[:,0].min(), " ends:", t[:,9].max()-t[:,0].min())
