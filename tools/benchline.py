#!/usr/bin/env python3
"""stdin (or the file named) = what bench.py printed -> ONE JSON object: the contract line's fields + every section.

    python bench.py ... | python tools/benchline.py | python -c "import sys, json; r = json.load(sys.stdin); ..."

The sections come from the side file the run wrote (`sections.file` of the contract line, in the current directory: every digit and
level) when it is there, else from the brief `#section` copies on stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchkit.line import collect, strict_loads   # noqa: E402

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line, full = collect(text)
side = (line.get("sections") or {}).get("file")
if side and os.path.exists(side):
    with open(side) as f:
        whole = strict_loads(f.read())
    if abs(whole.get("value", 0) - line["value"]) <= 1e-6 * abs(line["value"]):     # (the file of THIS run, not a stale one)
        full = dict(line, **{k: whole[k] for k in line["sections"]["names"] if k in whole})
print(json.dumps(full))
