"""tests/golden/matvec.json: the digests hehub ITSELF prints for examples/diag_matvec.cpp (the diagonal loop of matrix_vector_mul_short,
src/circuits/linear_algebra.h:104-136, compiled against hehub's own headers and linked with hehub alone: make -C oracle ref_matvec ->
oracle/_ref/ref_matvec_cpu), for the cases tests/test_matvec.py runs through the MI355X layer.
Run in the container that has /root/reference:   python tests/golden/make_matvec.py"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_matvec_cpu")
# (logN, L, width, mode)
CASES = [(11, 3, 4, "short"), (12, 4, 8, "short"), (13, 6, 20, "short"), (12, 2, 2, "short"), (5, 2, 3, "short"), (14, 5, 6, "short"),
         (15, 10, 16, "short"), (11, 3, 4, "full"), (13, 6, 7, "full"), (15, 10, 5, "full")]


def run(binary, case, env=None, reps=1):
    out = subprocess.run([binary] + [str(a) for a in case] + [str(reps)], capture_output=True, text=True, timeout=1800,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (case, out.stdout[-2000:], out.stderr[-2000:])
    f = {m.group(1): m.group(2) for m in re.finditer(r"([\w-]+) digest (\w+)", out.stdout)}
    ms = {m.group(1): float(m.group(2)) for m in re.finditer(r"([\w-]+) ([\d.]+) ms per product vector", out.stdout)}
    return f, ms, out.stdout


if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_matvec"], check=True)
    table = {" ".join(str(a) for a in c): run(REF, c)[0]["loop"] for c in CASES}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "matvec.json"), "w") as f:
        json.dump({"_comment": "hehub's own digests of examples/diag_matvec.cpp (tests/golden/make_matvec.py); key = logN L width mode",
                   "digests": table}, f, indent=1)
    print(len(table), "cases")
