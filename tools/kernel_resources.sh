#!/bin/bash
# VGPRs / spills / LDS / occupancy of every kernel in one source file: tools/kernel_resources.sh hehub_amd/csrc/hp_ntt_fast.hip [filter]
F=$1; PAT=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $F -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /VGPRs Spill:/{sp=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{oc=$(NF-1)} /LDS Size/{print name, "vgpr="v, "spill="sp, "scratch="sc, "occ="oc, "lds="$(NF-1)}' \
 | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//' | c++filt | grep -E "$PAT"
