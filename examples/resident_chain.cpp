// resident_chain.cpp -- a program written against hehub's PUBLIC API only (no hp_* call, no device pointer): the accumulate
// loop of hehub's examples/ckks_example.cpp:15-26 (ct_squared = ckks::mult(ct, ct, relin_key); ct_sum = ckks::add(ct_sum,
// ct_squared)), the rotation loop of hehub's bench/benchmarks.cpp:31-35 (ckks::rotate(ct, rot_key, 1)), then
// rescale_inplace and a look at the words.  Ciphertext and key words are synthetic (splitmix64): the program checks
// RING ARITHMETIC, word for word, not noise; it prints an FNV-1a-64 digest of every word of the results, which must be the
// same for
//     (a) hehub itself on the CPU                      make -C oracle ref_chain   -> oracle/_ref/ref_chain_cpu
//     (b) hehub's headers + the binding                (same target)              -> oracle/_ref/ref_chain_amd
//     (c) the own mirror of the interface (hehub.hpp)  tests/test_host_residency.py builds examples/resident_chain
// and, for (b) / (c), how many bytes crossed PCIe: with device-resident operands every input ciphertext and key goes up
// once and only the words the program finally reads come back.
//
//   resident_chain [logN=15] [L=10] [iterations=8]
#ifdef CHAIN_REFERENCE_HEADERS
#include "fhe/ckks/ckks.h"
#include "fhe/common/permutation.h"
#include "fhe/primitives/keys.h"
#else
#include "hehub.hpp"
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace hehub;

static u64 sm_state;
static u64 splitmix() {
    u64 z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// first primes of hehub's 50-bit and 40-bit rows (primelists.cpp:131, :85-86): the chain of ckks::create_params(32768, {50, 40 x 9}, 50, .)
static const u64 P50[] = {1125899904679937ull, 1125899903827969ull};
static const u64 P40[] = {1099510054913ull, 1099507695617ull, 1099506515969ull, 1099504549889ull, 1099503894529ull,
                          1099503370241ull, 1099502714881ull, 1099502518273ull, 1099501731841ull, 1099500814337ull};

static RnsPolynomial random_poly(size_t n, const std::vector<u64> &moduli) {
    RnsPolynomial p(n, moduli.size(), moduli);
    for (size_t k = 0; k < moduli.size(); k++)
        for (size_t i = 0; i < n; i++) p[(int)k][i] = splitmix() % moduli[k];
    p.rep_form = PolyRepForm::value;
    return p;
}

static RlweKsk random_key(size_t n, size_t L, const std::vector<u64> &mext) {
    RlweKsk key;
    for (size_t j = 0; j < L; j++) key.push_back(RlweCt{random_poly(n, mext), random_poly(n, mext)});
    return key;
}

static u64 fnv(u64 h, const RnsPolynomial &p) {
    for (size_t k = 0; k < p.component_count(); k++)
        for (size_t i = 0; i < p.dimension(); i++) {
            u64 w = p[(int)k][i];
            for (int b = 0; b < 8; b++) { h ^= (w >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
        }
    return h;
}

static u64 peek(const RnsPolynomial &p) { return p[0][0]; }   // (a const look: both copies of the polynomial stay current)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const size_t logn = argc > 1 ? std::atoi(argv[1]) : 15, L = argc > 2 ? std::atoi(argv[2]) : 10, iters = argc > 3 ? std::atoi(argv[3]) : 8;
    if (L < 2 || L > 10 || logn < 1 || logn > 15) { std::fprintf(stderr, "usage: resident_chain [logN<=15] [2<=L<=10] [iterations]\n"); return 2; }
    const size_t n = (size_t)1 << logn;
    std::vector<u64> q{P50[1]};
    for (size_t k = 1; k < L; k++) q.push_back(P40[k - 1]);
    std::vector<u64> mext(q);
    mext.push_back(P50[0]);
    sm_state = 2024;
    CkksCt ct1(RlweCt{random_poly(n, q), random_poly(n, q)}), ct2(RlweCt{random_poly(n, q), random_poly(n, q)});
    ct1.scaling_factor = ct2.scaling_factor = 1099511627776.0;   // 2^40
    RlweKsk relin_key = random_key(n, L, mext), rot_key = random_key(n, L, mext);

    // --- examples/ckks_example.cpp:15-26: multiply, accumulate ---------------------------------------------------------
    CkksCt ct_sum;
    double t0 = now_ms(), t_first = 0;
    for (size_t i = 0; i < iters; i++) {
        auto ct_prod = ckks::mult(ct1, ct2, relin_key);
        if (i == 0) {
            ct_sum = ct_prod;
            t_first = now_ms() - t0;   // includes every one-time cost: table build, upload of operands and key
            t0 = now_ms();
        } else {
            ct_sum = ckks::add(ct_sum, ct_prod);
        }
    }
    // --- bench/benchmarks.cpp:31-35: rotate --------------------------------------------------------------------------------
    const u64 probe_sum = peek(ct_sum[1]);   // looking at one word waits for the chain (and fetches that polynomial)
    const double t_chain_end = now_ms();
    CkksCt ct_rot = ckks::rotate(ct1, rot_key, 1);
    const u64 probe_rot0 = peek(ct_rot[1]);
    const double t1 = now_ms();
    for (size_t i = 1; i < iters; i++) ct_rot = ckks::rotate(ct_rot, rot_key, 1);
    const u64 probe_rot = peek(ct_rot[1]);
    const double t2 = now_ms();
    ckks::rescale_inplace(ct_sum);
    // --- the words are looked at only here ---------------------------------------------------------------------------------
    u64 h = 0xcbf29ce484222325ull;
    h = fnv(fnv(h, ct_sum[0]), ct_sum[1]);
    h = fnv(fnv(h, ct_rot[0]), ct_rot[1]);
    h ^= probe_sum ^ probe_rot0 ^ probe_rot;
    std::printf("shape N=%zu L=%zu iterations=%zu\n", n, L, iters);
    std::printf("digest %016llx\n", (unsigned long long)h);
    std::printf("first mult (one-time costs included) %.3f ms\n", t_first);
    if (iters > 1) {
        std::printf("mult+add per iteration %.3f ms\n", (t_chain_end - t0) / (double)(iters - 1));
        std::printf("rotate per iteration %.3f ms\n", (t2 - t1) / (double)(iters - 1));
    }
#ifndef CHAIN_REFERENCE_HEADERS
    const auto st = amd::transfer_stats();
    // read back: a look at ONE limb downloads that limb, a second limb the whole polynomial -- the four polynomials of the two results
    // (their first limb, then all: 2 (1 + L-1) + 2 (1 + L) limbs; a polynomial of one limb once) + the first limbs of the two probed polynomials that were replaced
    // afterwards (ct_sum[1] before the rescale, the first ct_rot[1]); the third probe is the first limb of the final ct_rot[1]
    const double in_mib = (double)(2 * 2 * L + 2 * (2 * L * (L + 1))) * n * 8 / 1048576.0, out_mib = (double)(2 * ((L - 1 == 1) ? 1 : L) + 2 * (1 + L) + 2) * n * 8 / 1048576.0;
    std::printf("pcie to_device %.2f MiB in %llu copies (operands + keys once = %.2f MiB)\n", st.h2d_bytes / 1048576.0, st.h2d_copies, in_mib);
    std::printf("pcie to_host %.2f MiB in %llu copies (the two results + the probed first limbs = %.2f MiB)\n", st.d2h_bytes / 1048576.0, st.d2h_copies, out_mib);
    std::printf("engine_calls %llu\n", st.engine_calls);
#endif
    return 0;
}
