# per-round phase stamps of a long forward launch (needs the -DHP_TRACE variant "trace"): how the rounds of a launch evolve from the
# lock-step start (all CUs load, compute, store together) to the drifted steady state
import ctypes as C, numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ.setdefault("HEHUB_AMD_LIB",os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so"))
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e=Engine(0)
lib=capi.load()
lib.hp_debug_trace.argtypes=[C.c_void_p,C.c_size_t]; lib.hp_debug_trace.restype=C.c_int
mods=P.C3_Q; n=1<<15; L=len(mods); B=int(sys.argv[1]) if len(sys.argv)>1 else 2560
x=torch.randint(0,1<<40,(B,L,n),dtype=torch.int64,device="cuda")
for _ in range(3): e.ntt_(mods,x)
torch.cuda.synchronize()
buf=np.zeros(4096*12,dtype=np.uint64)
lib.hp_debug_trace(buf.ctypes.data_as(C.c_void_p),buf.size)
W=B*L
t=buf.reshape(-1,12)[:2*((W+15)//16)].astype(np.int64)
w0=t[0::2]; wl=t[1::2]           # wave 0 / last wave of every 16th workgroup, in blockIdx order
t0=w0[:,11].min()
names="load,passA,exch1,passB,exch2,passC,fold,exch3,store".split(",")
print("round  start_us(entry of wave0, median)  wave0: load passA exch1(wait)  total | lastwave: load total | wg lifetime")
for r in range(0, W//256, 1):
    sl=slice(r*16,(r+1)*16)
    a=w0[sl]; b=wl[sl]
    if len(a)==0: break
    d=np.diff(a[:,:10],axis=1); db=np.diff(b[:,:10],axis=1)
    life=b[:,9]-a[:,11]
    if r<24 or r%10==0 or r==W//256-1:
        print(f"{r:3d}  {np.median(a[:,11]-t0)/2200:9.1f}   {np.median(d[:,0]):7.0f} {np.median(d[:,1]):7.0f} {np.median(d[:,2]):7.0f}  {np.median(a[:,9]-a[:,0]):7.0f} | {np.median(db[:,0]):7.0f} {np.median(b[:,9]-b[:,0]):7.0f} | {np.median(life):7.0f}")

