#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/prof_round.sh:  python tools/traffic_from_pmc.py <tag> [dir = profiles]
HBM bytes per limb transform = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / limbs of the dispatch (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads; both counters in KiB per dispatch), VALUBusy =
4 x SQ_ACTIVE_INST_VALU / 32 SIMDs per counter instance / GRBM_GUI_ACTIVE, VALU instructions per wave = SQ_INSTS_VALU / SQ_WAVES."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles"))
# summary file -> (kernel regex, entry name, limbs per dispatch)
SPECS = [("ntt15", r"k_ntt_fwdILi15", "k_ntt_fwd_logn15", 256 * 11), ("intt15", r"k_ntt_invILi15", "k_ntt_inv_logn15", 256 * 11),
         ("ntt", r"k_ntt_fwdILi14", "k_ntt_fwd_logn14", 1024 * 4), ("intt", r"k_ntt_invILi14", "k_ntt_inv_logn14", 1024 * 4),
         ("ntt12", r"k_ntt_fwdILi12", "k_ntt_fwd_logn12", 3724 * 11), ("intt12", r"k_ntt_invILi12", "k_ntt_inv_logn12", 3724 * 11),
         ("ckks", r"k_ntt_fwdILi15", "k_ntt_fwd_logn15_spread", 64 * 100), ("bgv", r"k_ntt_fwdILi13", "k_ntt_fwd_logn13_spread", 128 * 36),
         # parity level A (hp_ntt_a.hip): the same launches on the FP64 residue kernels
         ("ckks_a", r"k_ntt_fwd_aILi15", "k_ntt_fwd_a_logn15_spread", 64 * 100), ("bgv_a", r"k_ntt_fwd_aILi13", "k_ntt_fwd_a_logn13_spread", 128 * 36)]
SPECS = [s for s in SPECS if not os.environ.get("TRAFFIC_ONLY") or s[0] in os.environ["TRAFFIC_ONLY"].split(",")]
out_path = os.path.join(ROOT, "profiles", "traffic.json")
tr = json.load(open(out_path)) if os.path.exists(out_path) else {}
for name, kre, entry, limbs in SPECS:
    f = os.path.join(d, f"{tag}_pmc_{name}_summary.txt")
    if not os.path.exists(f):
        f = os.path.join(d, f"pmc_{tag}_{name}_summary.txt")
    if not os.path.exists(f):
        continue
    txt = open(f).read()
    def g(counter):
        m = re.search(kre + r"\S*\s+" + counter + r"\s+n=\d+\s+avg=(\S+)", txt)
        return float(m.group(1)) if m else None
    fetch, write = g("FETCH_SIZE"), g("WRITE_SIZE")
    if fetch is None or write is None:
        continue
    logn = int(re.search(r"logn(\d+)", entry).group(1))
    e = {"limbs_per_dispatch": limbs, "fetch_size_kb": fetch, "write_size_kb": write,
         "bytes_per_limb": round((2 * fetch + write) * 1024 / limbs), "algorithmic_bytes_per_limb": 16 << logn,
         "source": f"round {int(tag[1:3])} (profiles/{tag}_pmc_{name}_summary.txt)"}
    valu, gui, waves, insts = g("SQ_ACTIVE_INST_VALU"), g("GRBM_GUI_ACTIVE"), g("SQ_WAVES"), g("SQ_INSTS_VALU")
    if valu and gui:
        e["valu_busy"] = round(4 * valu / 32 / gui, 3)
    if waves and insts:
        e["valu_insts_per_wave"] = round(insts / waves)
    if valu and insts:
        e["valu_cycles_per_inst"] = round(4 * valu / insts, 3)   # SIMD cycles one VALU instruction of this kernel's mix occupies (4 = full rate)
    tr[entry] = e
    print(entry, e)
json.dump(tr, open(out_path, "w"), indent=1)
