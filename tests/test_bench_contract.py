"""bench.py prints ONE JSON line carrying every field of the driver's contract (small batch so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ckks", "ntt"])
def test_bench_line_has_the_contract_fields(workload):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--batch", "8", "--steps", "2",
                          "--warmup", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    r = json.loads(lines[0])
    assert REQUIRED <= set(r), REQUIRED - set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["dtype"] == "u64" and r["data"] == "synthetic" and r["vs_baseline"] is None and r["scaling"] == "weak"
    assert "workload" in r["config"] and r["value"] > 0 and r["ms_per_step"] > 0
    roof = r["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and roof["bound"] == "hbm" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    cpu = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] in ("reference", "port") and cpu["cores"] == 1
