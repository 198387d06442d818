#!/usr/bin/env python3
"""stdin (or the file named) = what bench.py printed -> ONE JSON object: the contract line's fields + every `#section` line.

    python bench.py ... | python tools/benchline.py | python -c "import sys, json; r = json.load(sys.stdin); ..."
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchkit.line import collect   # noqa: E402

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
_, full = collect(text)
print(json.dumps(full))
