#!/bin/bash
# Experiment: the workgroups of the FIRST dispatch round of a tiled transform launch start phase-shifted (blockIdx < 256: sleep
# ((blockIdx >> 3) % PHASES) * TICKS shader clocks), so that the CUs do not load / compute / store in lockstep during the first
# rounds of a launch (short launches -- 10-20 rounds -- cost 50-55 us per round against 43 in steady state).
#   tools/variants/mk_stagger.sh <name> <phases> <ticks>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; PH=$2; TK=$3
D=$R/hehub_amd/lib_variants/src_$NAME
rm -rf $D; mkdir -p $D/hehub_amd $D/obj
cp -r $R/hehub_amd/csrc $D/hehub_amd/csrc; cp -r $R/include $D/include
python3 - "$D/hehub_amd/csrc/hp_ntt_fast.hip" $PH $TK <<'PY'
import sys
p, ph, tk = sys.argv[1], sys.argv[2], sys.argv[3]
s = open(p).read()
code = f"""    if (blockIdx.x < 256u) {{
        const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)((blockIdx.x >> 3) % {ph}u) * {tk}ull;
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(32);
    }}
"""
a = "    const u32 w = hp_xcd_remap(blockIdx.x, job.W);\n    HpItem it;\n    if (!hp_decode_item(job, w, it)) return;\n    // the limb's constants"
assert a in s
s = s.replace(a, code + a, 1)
c = "    const u32 sub = threadIdx.x / G::T, tid = threadIdx.x % G::T;"
assert c in s
s = s.replace(c, code + c, 1)
open(p, "w").write(s)
PY
cd $D/hehub_amd/csrc
(for f in *.cpp *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $f -o $D/obj/${f%.*}.o 2>/dev/null & done; wait) > /dev/null 2>&1
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so $D/obj/*.o -Wl,-rpath,/opt/rocm/lib -lpthread
rm -rf $D
echo $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so
