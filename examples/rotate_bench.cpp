// rotate_bench.cpp -- hehub's OWN benchmark (bench/benchmarks.cpp:21-37) as a program over hehub's public API that both sides can run:
// `ckks::rotate(ct, rot_key, 1)`, one ciphertext per call, at the benchmark's four parameter sets -- N = 2^12 .. 2^15 with the modulus
// chains ckks::create_params(N, scaling_bits) draws (basics.cpp:40-66: 2 x 36, 44 + 3 x 43, 51 + 7 x 48, 15 x 55 bits, plus the
// additional modulus).  hehub's program encrypts a vector first; here ciphertext and key words are synthetic (splitmix64) -- the
// rotation does the same arithmetic on them, and the result can be compared WORD FOR WORD: the program prints an FNV-1a-64 digest of
// every rotated ciphertext, which must be the same for
//     (a) hehub itself on the CPU                      make -C oracle ref_rotbench -> oracle/_ref/ref_rotbench_cpu   (also checks the
//                                                      chains below against create_params itself)
//     (b) the own mirror of the interface (hehub.hpp)  hehub_amd.build.build_example("rotate_bench")  (tests/test_rotate_bench.py)
// and the time per rotation two ways: with a look at a word of every result before the next call (hehub's calls are synchronous:
// this is the like-for-like latency) and back to back (independent calls overlap over the layer's lanes; with HEHUB_AMD_DEFER=1
// they are recorded and run as one batch).
//
//   rotate_bench [reps=20] [only_logn=0 (all)]
#ifdef CHAIN_REFERENCE_HEADERS
#include "fhe/ckks/ckks.h"
#include "fhe/common/permutation.h"
#include "fhe/primitives/keys.h"
#else
#include "hehub.hpp"
#include "hehub_amd_ext.hpp"
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
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

// what ckks::create_params(N, scaling_bits) returns for the benchmark's four pairs (values printed by hehub itself; build (a) checks them)
struct ParamSet {
    size_t logn, scaling_bits;
    u64 additional;
    std::vector<u64> moduli;
};
static const ParamSet SETS[] = {
    {12, 36, 68718428161ull, {68714954753ull, 68713512961ull}},
    {13, 43, 17592182833153ull, {17592182243329ull, 8796090597377ull, 8796090007553ull, 8796087582721ull}},
    {14, 48, 2251799813554177ull,
     {2251799811391489ull, 281474976317441ull, 281474975662081ull, 281474974482433ull, 281474972188673ull, 281474971926529ull, 281474971533313ull,
      281474966880257ull}},
    {15, 55, 36028797017456641ull,
     {36028797014704129ull, 36028797014573057ull, 36028797014376449ull, 36028797013327873ull, 36028797013000193ull, 36028797012606977ull,
      36028797010444289ull, 36028797009985537ull, 36028797005856769ull, 36028797005529089ull, 36028797005135873ull, 36028797003694081ull,
      36028797003563009ull, 36028797001138177ull, 36028796998844417ull}},
};

static RnsPolynomial random_poly(size_t n, const std::vector<u64> &moduli) {
    RnsPolynomial p(n, moduli.size(), moduli);
    for (size_t k = 0; k < moduli.size(); k++)
        for (size_t i = 0; i < n; i++) p[(int)k][i] = splitmix() % moduli[k];
    p.rep_form = PolyRepForm::value;
    return p;
}

static u64 fnv(u64 h, const RnsPolynomial &p) {
    for (size_t k = 0; k < p.component_count(); k++) {
        const auto &limb = p[(int)k];
        for (size_t i = 0; i < p.dimension(); i++) {
            u64 w = limb[i];
            for (int b = 0; b < 8; b++) { h ^= (w >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
        }
    }
    return h;
}

static u64 peek(const RnsPolynomial &p) { return p[0][0]; }   // a const look at one word: the result has been computed

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const size_t reps = argc > 1 ? std::atoi(argv[1]) : 20, only = argc > 2 ? std::atoi(argv[2]) : 0;
    if (reps < 1) { std::fprintf(stderr, "usage: rotate_bench [reps>=1] [only_logn]\n"); return 2; }
    for (const ParamSet &ps : SETS) {
        if (only && ps.logn != only) continue;
        const size_t n = (size_t)1 << ps.logn, L = ps.moduli.size();
#ifdef CHAIN_REFERENCE_HEADERS
        {   // the chains above are hehub's: bench/benchmarks.cpp:24
            auto params = ckks::create_params(n, ps.scaling_bits);
            if (params.moduli != ps.moduli || params.additional_mod != ps.additional) { std::fprintf(stderr, "parameter table differs from create_params\n"); return 3; }
        }
#endif
        std::vector<u64> mext(ps.moduli);
        mext.push_back(ps.additional);
        sm_state = 7000 + ps.logn;
        CkksCt ct(RlweCt{random_poly(n, ps.moduli), random_poly(n, ps.moduli)});
        ct.scaling_factor = std::pow(2.0, (double)ps.scaling_bits);
        RlweKsk rot_key;
        for (size_t j = 0; j < L; j++) rot_key.push_back(RlweCt{random_poly(n, mext), random_poly(n, mext)});
        // warm: tables, the upload of the operands and of the key
        CkksCt warm = ckks::rotate(ct, rot_key, 1);
        u64 h = fnv(fnv(0xcbf29ce484222325ull, warm[0]), warm[1]);
        // two passes, the better one counts: the first also pays what is paid once (a lane's first call sizes its workspace, the first
        // batch of a recorded run sizes lane 0's)
        const u64 probe = peek(warm[1]);
        size_t differing = 0;   // (every rotation is the same rotation: the digest does not depend on reps)
        double ms_sync = 1e30, ms_b2b = 1e30;
        std::vector<CkksCt> keep;
        for (int pass = 0; pass < 2; pass++) {
            // hehub's loop: independent rotations of one ciphertext, each result looked at before the next call
            double t0 = now_ms();
            for (size_t r = 0; r < reps; r++) {
                CkksCt ct_rotated = ckks::rotate(ct, rot_key, 1);
                if (peek(ct_rotated[1]) != probe) differing++;
            }
            ms_sync = std::min(ms_sync, (now_ms() - t0) / (double)reps);
            // the same calls back to back (the results are kept, nobody looks until the end)
            keep.clear();
            keep.reserve(reps);
            t0 = now_ms();
            for (size_t r = 0; r < reps; r++) keep.push_back(ckks::rotate(ct, rot_key, 1));
#ifndef CHAIN_REFERENCE_HEADERS
            amd::synchronize();
#endif
            ms_b2b = std::min(ms_b2b, (now_ms() - t0) / (double)reps);
#ifdef CHAIN_REFERENCE_HEADERS
            break;   // (hehub on the CPU has nothing to warm beyond the first call)
#endif
        }
        h = fnv(fnv(h, keep.back()[0]), keep.back()[1]);
        h ^= peek(keep.front()[1]) + differing;
        std::printf("CKKS rotation / N=%zu / scaling=2^%zu / L=%zu: %.4f ms per rotation (a look after every call), %.4f ms back to back; digest %016llx\n",
                    n, ps.scaling_bits, L, ms_sync, ms_b2b, (unsigned long long)h);
    }
#ifndef CHAIN_REFERENCE_HEADERS
    const auto st = amd::transfer_stats();
    std::printf("layer: lanes %d deferred %d engine_calls %llu deferred_calls %llu deferred_groups %llu\n", amd::lanes(), (int)amd::deferred(),
                st.engine_calls, st.deferred_calls, st.deferred_groups);
#endif
    return 0;
}
