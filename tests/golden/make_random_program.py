"""tests/golden/random_program.json: the digests hehub ITSELF prints for examples/random_program.cpp (compiled against hehub's own
headers, linked with hehub alone: make -C oracle ref_randprog -> oracle/_ref/ref_randprog_cpu), for the cases tests/test_random_program.py
runs through the MI355X layer.  Run in the container that has /root/reference:   python tests/golden/make_random_program.py"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_randprog_cpu")
# (logN, L, pool, ops, seed, bgv)
CASES = ([(12, 4, 12, 400, s, 0) for s in range(1, 9)] + [(13, 6, 10, 250, s, 0) for s in (11, 12, 13)] + [(11, 2, 6, 300, s, 0) for s in (21, 22)]
         + [(10, 3, 16, 600, 31, 0), (14, 5, 8, 120, 41, 0)] + [(12, 4, 12, 400, s, 1) for s in (51, 52, 53)] + [(13, 6, 8, 200, 61, 1)])


def digest(binary, case, env=None):
    out = subprocess.run([binary] + [str(a) for a in case], capture_output=True, text=True, timeout=1800, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (case, out.stdout[-2000:], out.stderr[-2000:])
    return re.search(r"program digest (\w+)", out.stdout).group(1), out.stdout


if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_randprog", "ref_randprog_amd"], check=True)
    table = {" ".join(str(a) for a in c): digest(REF, c)[0] for c in CASES}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "random_program.json"), "w") as f:
        json.dump({"_comment": "hehub's own digests of examples/random_program.cpp (tests/golden/make_random_program.py); key = logN L pool ops seed bgv",
                   "digests": table}, f, indent=1)
    print(len(table), "cases")
