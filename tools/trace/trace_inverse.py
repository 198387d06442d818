# phase stamps of the inverse transform (needs the -DHP_TRACE variant "trace": tools/build_variant.sh trace -DHP_TRACE)
import ctypes as C, numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
os.environ.setdefault("HEHUB_AMD_LIB",os.path.abspath("hehub_amd/lib_variants/libhehub_amd_trace.so"))
import torch, params as P
from hehub_amd.engine import Engine
from hehub_amd import capi
e=Engine(0)
lib=capi.load()
lib.hp_debug_trace.argtypes=[C.c_void_p,C.c_size_t]; lib.hp_debug_trace.restype=C.c_int
names="load,exch0,passA,exch1,passB,exch2,passC,scale,store".split(",")
def dump(W,label):
    nrec=((W+15)//16)*2
    buf=np.zeros(4096*12,dtype=np.uint64)
    lib.hp_debug_trace(buf.ctypes.data_as(C.c_void_p),buf.size)
    t=buf.reshape(-1,12)[:min(nrec,4096)].astype(np.int64)
    d=np.diff(t[:,:10],axis=1)
    for w,lab in ((0,"wave0"),(1,"lastwave")):
        dd=d[w::2]
        print(label,f"W={W:5d}",lab," ".join(f"{nm}={np.median(dd[:,i]):.0f}" for i,nm in enumerate(names)), "total=%.0f"%np.median(t[w::2,9]-t[w::2,0]))
    # start of the workgroup relative to the end of the workgroup that ran before it on the same CU is not visible here; the
    # spread between wave 0 and the last wave at entry is:
    print(label, "mark0 skew lastwave-wave0 median", np.median(t[1::2,0]-t[0::2,0]), " end skew", np.median(t[1::2,9]-t[0::2,9]),
          " kernel-entry skew", np.median(t[1::2,11]-t[0::2,11]), " entry->mark0 wave0", np.median(t[0::2,0]-t[0::2,11]), " lastwave", np.median(t[1::2,0]-t[1::2,11]))
for logn in (15,14):
    mods=P.C3_Q; n=1<<logn; L=len(mods)
    for B in ((512, 2560) if logn == 15 else (1024,)):
        x=torch.randint(0,1<<40,(B,L,n),dtype=torch.int64,device="cuda")
        for _ in range(2): e.intt_(mods,x)
        torch.cuda.synchronize(); dump(B*L,f"inv logn={logn} B={B}")
        for _ in range(2): e.ntt_(mods,x)
        torch.cuda.synchronize()
        names_f="load,passA,exch1,passB,exch2,passC,fold,exch3,store".split(",")
        nm=names; names=names_f; dump(B*L,f"fwd logn={logn} B={B}"); names=nm
        del x
