"""Parameter sets shared by tests, fixtures and bench.

Prime values are DATA taken from the reference's table
(src/fhe/common/primelists.cpp: 40-bit row :85-86, 50-bit row :131) and from
its tests (tests/ntt_t.cpp:21-23,94-96, tests/mod_arith_t.cpp:8,36,
tests/bgv_t.cpp:28).
"""

# first primes of the 50-bit row; create_params draws them in this order
P50 = [1125899904679937, 1125899903827969, 1125899903500289, 1125899903107073]
# first primes of the 40-bit row
P40 = [1099510054913, 1099507695617, 1099506515969, 1099504549889, 1099503894529,
       1099503370241, 1099502714881, 1099502518273, 1099501731841, 1099500814337]

# moduli exercised by the reference's NTT tests
NTT_TEST_Q = [65537, 260898817, 35184358850561, 36028796997599233, 576460752272228353]
# moduli of tests/mod_arith_t.cpp "batched barrett"
BARRETT_TEST_Q = [65537, 33333333, 777777777777777, 1234567890111111111]
MULMOD_TEST_Q = 1234567890111111111

# C1: bench/ntt_bm.cpp:9
C1_Q = 576460752272228353
C1_LOGN = 12

# C2: N=16384, first four 50-bit primes, batch 1024
C2_LOGN = 14
C2_MODULI = P50[:4]
C2_BATCH = 1024

# C3/C4: ckks::create_params(32768, {50, 40 x 9}, 50, 2^40): p (additional) is
# drawn first from the 50-bit row, then q0 from the same row, then nine 40-bit
C3_LOGN = 15
C3_P = P50[0]
C3_Q = [P50[1]] + P40[:9]
C3_MODULI_EXT = C3_Q + [C3_P]
C3_BATCH = 256

# C5: create_params(8192, {40 x 7}): q0..q5 then p
C5_LOGN = 13
C5_Q = P40[:6]
C5_P = P40[6]
C5_MODULI_EXT = C5_Q + [C5_P]
C5_T = 65537
C5_BATCH = 4096


def edge_case(rng, logn, moduli):
    """Inputs for the functions either side of the path, from one splitmix64 stream:
    noise int64[N] in [-19, 19] (6 sigma of the reference's sigma = 3.2), c1 / sk uniform NTT-form words,
    pt uniform coefficient-form words."""
    import numpy as np

    n, L = 1 << logn, len(moduli)
    noise = (rng.words(n, 39).astype(np.int64) - 19)
    c1 = rng.poly((L, n), moduli)
    pt = rng.poly((L, n), moduli)
    sk = rng.poly((L, n), moduli)
    return noise, c1, pt, sk


def small_rns_poly(rng, n, moduli, bound=1000):
    """RNS limbs of a polynomial whose centred coefficients are in (-bound, bound): the small-coefficient case of
    rns_base_transform_to_single (rns_transform.cpp:45-82)."""
    import numpy as np

    v = rng.words(n, 2 * bound - 1).astype(np.int64) - (bound - 1)
    return np.stack([np.where(v >= 0, v, v + int(q)).astype(np.uint64) for q in moduli])


def ntt_primes(count, logn, bits, exclude=()):
    """the first `count` primes q = 1 (mod 2N) below 2^bits that are not in `exclude` (deterministic Miller-Rabin)"""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2; r += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    step, q, out = 2 << logn, ((1 << bits) // (2 << logn)) * (2 << logn) + 1, []
    while len(out) < count:
        q -= step
        if q not in exclude and is_prime(q):
            out.append(q)
    return out
