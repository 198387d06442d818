#!/bin/bash
# Per-file scheduler strategies: tools/build_variant2.sh <name> <strategy for hp_ntt_fast.hip|-> <strategy for hp_elem.hip|-> [more flags]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SN=$2; SE=$3; shift 3
D=$R/hehub_amd/lib_variants/obj_$NAME; mkdir -p $D
cd $R/hehub_amd/csrc
for f in hp_engine.cpp hp_tables.cpp hp_wire.cpp hp_elem.hip hp_hks.hip hp_ntt_generic.hip hp_ntt_fast.hip; do
  X=""
  if [ $f = hp_ntt_fast.hip ] && [ "$SN" != "-" ]; then X="-mllvm -amdgpu-sched-strategy=$SN"; fi
  if [ $f = hp_elem.hip ] && [ "$SE" != "-" ]; then X="-mllvm -amdgpu-sched-strategy=$SE"; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $X "$@" -x hip -c $f -o $D/${f%.*}.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so $D/*.o -Wl,-rpath,/opt/rocm/lib
rm -rf $D
echo $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so
