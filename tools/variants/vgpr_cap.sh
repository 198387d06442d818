#!/bin/bash
# Experiment: cap the VGPR allocation of the tiled transform kernels below the 128 of "4 waves per SIMD" so that one wave per
# SIMD of a lean HBM-bound kernel (k_tensor: 32 VGPRs) can be resident NEXT to a transform workgroup on the same CU, then run the
# mult pipeline as two software-pipelined sub-batches (HP_MULT_STREAMS=2).   tools/variants/vgpr_cap.sh <cap, e.g. 120>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
CAP=$1
P=$(mktemp)
cat > $P <<PATCH
PATCH
D=$R/hehub_amd/lib_variants/src_cap$CAP
rm -rf $D; mkdir -p $D/hehub_amd $D/obj
cp -r $R/hehub_amd/csrc $D/hehub_amd/csrc; cp -r $R/include $D/include
sed -i "s/__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW)/__global__ void __attribute__((amdgpu_num_vgpr($CAP))) __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW)/; s/__global__ void __launch_bounds__(InvGeo<LOGN>::TT, Geo<LOGN>::MINW)/__global__ void __attribute__((amdgpu_num_vgpr($CAP))) __launch_bounds__(InvGeo<LOGN>::TT, Geo<LOGN>::MINW)/" $D/hehub_amd/csrc/hp_ntt_fast.hip
cd $D/hehub_amd/csrc
for f in *.cpp *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $f -o $D/obj/${f%.*}.o 2>/dev/null & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/hehub_amd/lib_variants/libhehub_amd_cap$CAP.so $D/obj/*.o -Wl,-rpath,/opt/rocm/lib -lpthread
rm -rf $D $P
echo $R/hehub_amd/lib_variants/libhehub_amd_cap$CAP.so
