#!/bin/bash
# Build an experimental variant of the engine: tools/build_variant.sh <name> [-DFLAG ...]
# -> hehub_amd/lib_variants/libhehub_amd_<name>.so ; run with HEHUB_AMD_LIB=<that path>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/hehub_amd/lib_variants
cd $R/hehub_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -x hip hp_engine.cpp hp_tables.cpp hp_wire.cpp hp_elem.hip hp_hks.hip hp_ntt_generic.hip hp_ntt_fast.hip \
  -o $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so -Wl,-rpath,/opt/rocm/lib
echo $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so
