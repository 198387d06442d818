#!/usr/bin/env python3
"""profiles/traffic.json "step_<workload>_<level>" from the counter passes of tools/prof_step_traffic.sh:
    python tools/step_traffic_from_pmc.py <tag> [dir = profiles]
HBM bytes of one step = sum over every ENGINE kernel of the step of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 x dispatches per step
(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads; both counters in KiB per dispatch).  The
passes run `bench.py --workload <wl> --parity-level <lv> --steps 2 --warmup 1 --roofline-only`: 3 identical steps at the default batch;
torch's own kernels (input generation) are not part of a step and are left out."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles"))
STEPS = 3
SHAPES = {"ckks": {"N": 32768, "L": 10, "batch": 256}, "bgv": {"N": 8192, "L": 6, "batch": 512}}
ENGINE = re.compile(r"k_ntt|k_ks_|k_tensor|k_poly|k_gather|k_copy|k_drop|k_vec|k_reverse|k_host_rows")
out_path = os.path.join(ROOT, "profiles", "traffic.json")
tr = json.load(open(out_path)) if os.path.exists(out_path) else {}
for wl, shape in SHAPES.items():
    for lv, suf in (("B", ""), ("A", "_a")):
        f = os.path.join(d, f"{tag}_pmc_step_{wl}{suf}_summary.txt")
        if not os.path.exists(f):
            continue
        per = {}
        for line in open(f):
            m = re.match(r"(\S+)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+avg=(\S+)", line)
            if not m or not ENGINE.search(m.group(1)):
                continue
            name = re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_Z\d+", "", m.group(1))
            name = re.sub(r"\.kd$|\.\.\.$", "", name)
            e = per.setdefault(name, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches_per_step": 0.0})
            e[m.group(2)] = float(m.group(4)) * int(m.group(3)) / STEPS          # KiB per step of this kernel
            e["dispatches_per_step"] = int(m.group(3)) / STEPS
        if not per:
            continue
        by_kernel = {k: round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) for k, v in per.items()}
        ent = dict(shape, bytes_per_step=sum(by_kernel.values()), fetch_kb_per_step=round(sum(v["FETCH_SIZE"] for v in per.values())),
                   write_kb_per_step=round(sum(v["WRITE_SIZE"] for v in per.values())), by_kernel=by_kernel,
                   dispatches_per_step={k: v["dispatches_per_step"] for k, v in per.items()},
                   source=f"round {int(tag[1:3])} (profiles/{tag}_pmc_step_{wl}{suf}_summary.txt: 2 x FETCH_SIZE + WRITE_SIZE over every kernel of the step)")
        tr[f"step_{wl}_{lv}"] = ent
        print(f"step_{wl}_{lv}: {ent['bytes_per_step'] / 1e9:.3f} GB per step = {ent['bytes_per_step'] / shape['batch'] / 1e6:.1f} MB per op")
json.dump(tr, open(out_path, "w"), indent=1)
