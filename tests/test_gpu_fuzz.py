"""Randomised parity sweep (tools/fuzz_parity.py): random ring sizes, moduli sets, batch sizes and entry points --
transforms, coefficient-wise ops, gathers, CKKS / BGV pipelines, rotations, drops, encrypt / decrypt cores and the
limb-sharded composition -- every result compared word for word with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_are_bit_exact(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "15", str(seed)], capture_output=True,
                         text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-1000:])
    assert out.returncode == 0 and "fuzz ok" in out.stdout


@pytest.mark.gpu
def test_random_cases_at_parity_level_a():
    """the same sweep with the context at parity level A: pipelines against the oracle's words modulo q, residue transforms"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "15", "21", "levela"], capture_output=True,
                         text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-1000:])
    assert out.returncode == 0 and "fuzz ok" in out.stdout
