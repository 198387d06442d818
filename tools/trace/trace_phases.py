import ctypes as C, numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ["HEHUB_AMD_LIB"]=os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e=Engine(0)
logn,mods=15,P.C3_MODULI_EXT
n=1<<logn; L=len(mods)-1
lib=capi.load()
lib.hp_debug_trace.argtypes=[C.c_void_p,C.c_size_t]; lib.hp_debug_trace.restype=C.c_int
names=os.environ.get("TRACE_NAMES","load,passA,exch1,passB,exch2,passC,fold,exch3,store").split(",")
def dump(W,label):
    nrec=((W+15)//16)*2
    buf=np.zeros(4096*12,dtype=np.uint64)
    lib.hp_debug_trace(buf.ctypes.data_as(C.c_void_p),buf.size)
    t=buf.reshape(-1,12)[:min(nrec,4096)].astype(np.int64)
    d=np.diff(t[:,:10],axis=1)
    for w,lab in ((0,"wave0"),(1,"lastwave")):
        dd=d[w::2]
        print(label,f"W={W:5d}",lab," ".join(f"{nm}={np.median(dd[:,i]):.0f}" for i,nm in enumerate(names)), "total=%.0f"%np.median(t[w::2,9]-t[w::2,0]))
B=256
x=torch.randint(0,1<<40,(B,L+1,n),dtype=torch.int64,device="cuda")
for _ in range(2): e.ntt_(mods,x)
torch.cuda.synchronize(); dump(B*(L+1),"plain")
ct=torch.randint(0,1<<40,(B,2,L,n),dtype=torch.int64,device="cuda")
for _ in range(2): e.ckks_rescale(mods[:L],ct)
torch.cuda.synchronize(); dump(2*B*(L-1),"drop ")
