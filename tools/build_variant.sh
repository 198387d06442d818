#!/bin/bash
# Build an experimental variant of the engine WITHOUT touching the shipped sources:
#   tools/build_variant.sh <name> [--patch tools/variants/<x>.patch ...] [--sed <file>=<sed expression> ...] [--sched <file.hip>=<llvm strategy>] [hipcc flags ...]
# copies hehub_amd/csrc + include to hehub_amd/lib_variants/src_<name>, applies the patches there (patch -p1 from the
# repo root layout), compiles every source in parallel and links hehub_amd/lib_variants/libhehub_amd_<name>.so.
# Run with HEHUB_AMD_LIB=<that path>.  Experiments (ablations, A/B switches, -DHP_TRACE) live as patches / flags here.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
PATCHES=(); FLAGS=(); SEDS=(); declare -A SCHED
while [ $# -gt 0 ]; do
  case "$1" in
    --patch) PATCHES+=("$2"); shift 2;;
    --sched) SCHED[${2%%=*}]=${2#*=}; shift 2;;
    --sed) SEDS+=("$2"); shift 2;;
    *) FLAGS+=("$1"); shift;;
  esac
done
D=$R/hehub_amd/lib_variants/src_$NAME
rm -rf $D; mkdir -p $D/hehub_amd $D/obj
cp -r $R/hehub_amd/csrc $D/hehub_amd/csrc; cp -r $R/include $D/include
for p in "${PATCHES[@]}"; do (cd $D && patch -s -p1 < $R/$p); done
cd $D/hehub_amd/csrc
for e in "${SEDS[@]}"; do sed -i "${e#*=}" "${e%%=*}"; done
for f in *.cpp *.hip; do
  X=""; [ -n "${SCHED[$f]}" ] && X="-mllvm -amdgpu-sched-strategy=${SCHED[$f]}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $X "${FLAGS[@]}" -x hip -c $f -o $D/obj/${f%.*}.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so $D/obj/*.o -Wl,-rpath,/opt/rocm/lib
rm -rf $D
echo $R/hehub_amd/lib_variants/libhehub_amd_$NAME.so
