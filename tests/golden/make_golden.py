#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
It builds oracle/_ref/libhehub_ref.so (oracle/Makefile: the reference's own
sources compiled in place + oracle/ref_shim.cpp), evaluates every case of
tests/golden/cases.py through it, and writes the summaries.  The fixture holds
data only (inputs are regenerated from splitmix64 seeds; outputs are digests,
head/tail words and small full vectors).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from cases import run_cases  # noqa: E402
from oracle.pyoracle import Oracle, build  # noqa: E402

if __name__ == "__main__":
    build(ref=True)
    res = run_cases(Oracle("ref"), big=True)
    meta = {"generator": "tests/golden/make_golden.py",
            "source": "primihub/hehub reference compiled by oracle/Makefile (g++ -O2), via oracle/ref_shim.cpp",
            "digest": "FNV-1a-64 over little-endian raw u64 output words",
            "inputs": "splitmix64 streams, see tests/golden/cases.py"}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump({"meta": meta, "cases": res}, f, indent=0, sort_keys=True)
    print(f"wrote {len(res)} cases")

    # one object in the wire format (hehub_amd/csrc/hp_wire.cpp): the reference's ckks::mult + relinearize +
    # rescale_inplace result of the "n8" scheme case; pins both the byte layout and the words it carries
    from cases import wire_fixture_case  # noqa: E402
    from hehub_amd import wire  # noqa: E402

    mext, ct1, ct2, key = wire_fixture_case()
    out = Oracle("ref").ckks_mult(mext, ct1, ct2, key)
    blob = wire.pack(wire.CT, mext[:len(mext) - 2], out, rep_form=1, scalar=2.0 ** 30)
    with open(os.path.join(HERE, "ckks_mult_n8.hehubamd"), "wb") as f:
        f.write(blob)
    print(f"wrote ckks_mult_n8.hehubamd ({len(blob)} bytes)")
