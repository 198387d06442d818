"""tests/golden/rotate_bench.json: the digests hehub ITSELF prints for examples/rotate_bench.cpp (the loop of hehub's own benchmark,
bench/benchmarks.cpp:21-37, at its four parameter sets; compiled against hehub's own headers and linked with hehub alone:
make -C oracle ref_rotbench -> oracle/_ref/ref_rotbench_cpu, which also checks the program's modulus chains against
ckks::create_params).  Run in the container that has /root/reference:   python tests/golden/make_rotate_bench.py"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_rotbench_cpu")
LOGNS = (12, 13, 14, 15)


def run(binary, reps=1, only=0, env=None):
    """-> {logN: (digest, ms with a look after every call, ms back to back)}, stdout"""
    out = subprocess.run([binary, str(reps), str(only)], capture_output=True, text=True, timeout=1800, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rows = {}
    for m in re.finditer(r"N=(\d+) / scaling=2\^\d+ / L=\d+: ([\d.]+) ms per rotation \(a look after every call\), ([\d.]+) ms back to back; digest (\w+)",
                         out.stdout):
        rows[int(m.group(1)).bit_length() - 1] = (m.group(4), float(m.group(2)), float(m.group(3)))
    return rows, out.stdout


if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_rotbench"], check=True)
    rows, _ = run(REF)
    assert set(rows) == set(LOGNS)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rotate_bench.json"), "w") as f:
        json.dump({"_comment": "hehub's own digests of examples/rotate_bench.cpp (tests/golden/make_rotate_bench.py); key = logN",
                   "digests": {str(k): rows[k][0] for k in LOGNS}}, f, indent=1)
    print(len(rows), "parameter sets")
