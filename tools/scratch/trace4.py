import ctypes as C, numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ["HEHUB_AMD_LIB"]=os.path.abspath("hehub_amd/lib_variants/libhehub_amd_traceall.so")
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e=Engine(0)
logn,mods,B=15,P.C3_MODULI_EXT,256
n=1<<logn; L=len(mods); W=B*L
x=torch.randint(0,1<<40,(B,L,n),dtype=torch.int64,device="cuda")
for _ in range(2): e.ntt_(mods,x)
torch.cuda.synchronize()
lib=capi.load()
lib.hp_debug_trace.argtypes=[C.c_void_p,C.c_size_t]; lib.hp_debug_trace.restype=C.c_int
buf=np.zeros(2*W*12,dtype=np.uint64)
print("rc",lib.hp_debug_trace(buf.ctypes.data_as(C.c_void_p),buf.size))
t=buf.reshape(W,2,12)
w0=t[:,0,:]; w15=t[:,1,:]
hw=w0[:,10]; cu=((hw>>32)&0xFFFF) | ((hw&0xF)<<16)   # (se,sh,cu bits of HW_ID) + xcc
hwid=(hw>>32).astype(np.int64); xcc=(hw&0xFFFFFFFF).astype(np.int64)
cuid=((hwid>>8)&0xFF)*16+xcc   # cu_id+sh+se bits 8..15, plus xcc
start=w0[:,0].astype(np.int64); end=np.maximum(w0[:,9],w15[:,9]).astype(np.int64)
print("distinct CU ids:",len(np.unique(cuid)))
gaps=[];durs=[]
for c in np.unique(cuid):
    idx=np.where(cuid==c)[0]; o=np.argsort(start[idx]); s=start[idx][o]; en=end[idx][o]
    durs+=list(en-s); gaps+=list(s[1:]-en[:-1])
print("WGs per CU (median):",np.median([np.sum(cuid==c) for c in np.unique(cuid)]))
print("duration first-mark->last-mark: median %.0f  p90 %.0f"%(np.median(durs),np.percentile(durs,90)))
print("gap end(prev WG)->first mark(next WG) on same CU: median %.0f  p10 %.0f p90 %.0f"%(np.median(gaps),np.percentile(gaps,10),np.percentile(gaps,90)))
print("kernel span:",end.max()-start.min())
