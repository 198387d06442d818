"""splitmix64 input generator in plain numpy -- no checker library involved.

The same stream as oracle.pyoracle.SplitMix (tests/test_oracle_golden.py asserts it); it lives here so that
the golden-fixture test of the HIP engine (tests/test_gpu_golden.py) needs nothing from oracle/.
"""
import numpy as np


class SplitMix:
    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def words(self, n: int, q: int = 0) -> np.ndarray:
        gamma = np.uint64(0x9E3779B97F4A7C15)
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = np.uint64(self.state) + idx * gamma
            self.state = int(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        if q:
            z = z % np.uint64(q)
        return z

    def poly(self, shape, moduli) -> np.ndarray:
        """uniform words in [0,q_k) for an array [..., L, N]; one stream, limb-major."""
        shape = tuple(shape)
        L, n = shape[-2], shape[-1]
        out = np.empty(shape, dtype=np.uint64)
        flat = out.reshape(-1, L, n)
        for b in range(flat.shape[0]):
            for k in range(L):
                flat[b, k] = self.words(n, int(moduli[k]))
        return out
