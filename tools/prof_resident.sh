#!/bin/bash
# Device residency behind hehub's object API (GPU box): the same program (examples/resident_chain.cpp) as hehub on the CPU, as
# hehub's headers over the binding (with / without the opt-in caches) and as the own mirror.   tools/prof_resident.sh > out.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for a in "15 10 240" "13 6 240"; do   # (240 iterations: the one-time costs of every lane -- fork, scratch, pooled blocks -- are in the first few)
  echo "== own mirror (examples/resident_chain $a)"; HEHUB_AMD_DEFER=0 examples/resident_chain $a
  [ -x oracle/_ref/ref_chain_cpu ] && { echo "== hehub on the CPU (oracle/_ref/ref_chain_cpu $a)"; oracle/_ref/ref_chain_cpu $a; }
  [ -x oracle/_ref/ref_chain_amd ] && { echo "== hehub's headers over the binding, HEHUB_AMD_CT_CACHE=64 HEHUB_AMD_KEY_CACHE=4"
                                        HEHUB_AMD_CT_CACHE=64 HEHUB_AMD_KEY_CACHE=4 HEHUB_AMD_VERBOSE=1 oracle/_ref/ref_chain_amd $a 2>&1
                                        echo "== hehub's headers over the binding, HEHUB_AMD_KEY_CACHE=4 only"
                                        HEHUB_AMD_KEY_CACHE=4 HEHUB_AMD_VERBOSE=1 oracle/_ref/ref_chain_amd $a 2>&1
                                        echo "== hehub's headers over the binding, no caches, HEHUB_AMD_PIN_HOST=0 (round-3 transfers: one pageable copy per limb)"
                                        HEHUB_AMD_PIN_HOST=0 HEHUB_AMD_VERBOSE=1 oracle/_ref/ref_chain_amd $a 2>&1
                                        echo "== hehub's headers over the binding, no caches"; HEHUB_AMD_VERBOSE=1 oracle/_ref/ref_chain_amd $a 2>&1; }
done
