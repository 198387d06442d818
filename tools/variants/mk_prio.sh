#!/bin/bash
# Experiment variants (built from a patched copy): tools/variants/mk_prio.sh <name> <mode>
#   mode prio    : s_setprio 3 from kernel entry until the coefficient loads are issued (the preamble of the YOUNGEST waves is
#                  otherwise starved by the older waves' butterflies: tools/trace_phases.py shows 13 k cycles for it)
#   mode barrier : a workgroup barrier right after the loads are issued
#   mode both
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; MODE=$2; shift 2
D=$R/hehub_amd/lib_variants/src_$NAME
rm -rf $D; mkdir -p $D/hehub_amd $D/obj
cp -r $R/hehub_amd/csrc $D/hehub_amd/csrc; cp -r $R/include $D/include
python3 - "$D/hehub_amd/csrc/hp_ntt_fast.hip" "$MODE" <<'PY'
import sys
p, mode = sys.argv[1], sys.argv[2]
s = open(p).read()
pre = "    __builtin_amdgcn_s_setprio(3);\n" if mode in ("prio", "both") else ""
post = ("    __builtin_amdgcn_s_setprio(0);\n" if mode in ("prio", "both") else "") + ("    __syncthreads();\n" if mode in ("barrier", "both") else "")
# forward
a = "    const u32 w = hp_xcd_remap(blockIdx.x, job.W);\n    HpItem it;\n    if (!hp_decode_item(job, w, it)) return;\n    // the limb's constants"
assert a in s
s = s.replace(a, pre + a, 1)
b = "    load_flight<LOGN, LZ>(it.src, tid, x);\n"
assert b in s
s = s.replace(b, b + post, 1)
# inverse
c = "    const u32 sub = threadIdx.x / G::T, tid = threadIdx.x % G::T;"
assert c in s
s = s.replace(c, pre + c, 1)
d = "            x[2 * r + 1] = v.y;\n        }\n    }\n#pragma unroll\n    for (int i = 0; i < NSTG; ++i) {\n        const u32 e = threadIdx.x + (u32)i * TT;\n        if (e < 31u * 32u) lds_tw[e] = stg[i];"
assert d in s
s = s.replace(d, "            x[2 * r + 1] = v.y;\n        }\n    }\n" + post + d[len("            x[2 * r + 1] = v.y;\n        }\n    }\n"):], 1)
open(p, "w").write(s)
PY
cd $D/hehub_amd/csrc
(for f in *.cpp *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -x hip -c $f -o $D/obj/${f%.*}.o 2>/dev/null & done; wait) > /dev/null 2>&1
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so $D/obj/*.o -Wl,-rpath,/opt/rocm/lib -lpthread
rm -rf $D
echo $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so
