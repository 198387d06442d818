// diag_matvec.cpp -- the diagonal loop of hehub's only circuit-level caller of the key switch, matrix_vector_mul_short
// (src/circuits/linear_algebra.h:104-136), through hehub's one-ciphertext-per-call interface:
//
//   short    width <= slots / 2:  per diagonal i  mult_plain(rotating, diag[i]); accumulate; rotating = add(rotate(ct_vec, key[i+1]),
//            rotate(ct_vec, key[i+1 + slots - width]))            -- 2 (width - 1) INDEPENDENT rotations of one vector, no two under one key
//   full     width == slots:      rotating = rotate(rotating, key[1])  -- a dependent chain of width - 1 rotations under one key
//
// The loop below makes exactly the calls of linear_algebra.h:109-136 in its order; what is synthetic (splitmix64) are the WORDS: the
// ciphertext, the rotation keys and the encoded diagonals (coefficient form, as simd_encode returns them), so the program checks ring
// arithmetic word for word and needs no sampling: it prints an FNV-1a-64 digest over every word of the result, which must be the same
// in every mode AND for hehub itself on the CPU (make -C oracle ref_matvec -> oracle/_ref/ref_matvec_cpu: this file against hehub's own
// headers; ref_matvec_amd: the same over the binding).
//
// Own-mirror build (examples/diag_matvec): the loop runs eagerly (independent calls overlap over the layer's lanes), then in deferred
// mode (amd::set_deferred: the 2 (width - 1) rotations are recorded and run as ONE launch sequence with a key per ciphertext,
// hp_dev_ckks_rotate_many; the sums and plaintext products as batched calls), then "short" once more written with the batched form
// amd::rotate(cts, keys, steps) of hehub_amd_ext.hpp.
//
//   diag_matvec [logN=15] [L=10] [width=16] [mode=short|full] [reps=3]
#ifdef CHAIN_REFERENCE_HEADERS
#include "fhe/ckks/ckks.h"
#include "fhe/primitives/keys.h"
#else
#include "hehub.hpp"
#define MATVEC_AMD_EXT 1
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace hehub;
using ckks::CkksPt;

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

static RnsPolynomial random_poly(size_t n, const std::vector<u64> &moduli, PolyRepForm form) {
    RnsPolynomial p(n, moduli.size(), moduli);
    for (size_t k = 0; k < moduli.size(); k++) {
        auto &limb = p[(int)k];
        for (size_t i = 0; i < n; i++) limb[i] = splitmix() % moduli[k];
    }
    p.rep_form = form;
    return p;
}

static u64 digest(const CkksCt &ct) {
    u64 h = 0xcbf29ce484222325ull;
    for (int half = 0; half < 2; half++)
        for (size_t k = 0; k < ct[half].component_count(); k++) {
            const auto &limb = ct[half][(int)k];
            for (size_t i = 0; i < ct[half].dimension(); i++) {
                u64 w = limb[i];
                for (int b = 0; b < 8; b++) { h ^= (w >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
            }
        }
    return h;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void device_done() {
#ifdef MATVEC_AMD_EXT
    amd::synchronize();
#endif
}

// linear_algebra.h:104-136, call for call (the diagonals arrive encoded)
static CkksCt diag_loop(const std::vector<CkksPt> &diags, const CkksCt &ct_vec, const std::vector<RotKey> &rot_key_set, size_t slot_count,
                        bool full_width) {
    const size_t matrix_width = diags.size();
    auto ct_vec_rotating(ct_vec);
    CkksCt ct_accumulated;
    for (size_t i = 0; i < matrix_width; i++) {
        auto ct_prod_diag_vec = ckks::mult_plain(ct_vec_rotating, diags[i]);
        if (i == 0) {
            ct_accumulated = std::move(ct_prod_diag_vec);
        } else {
            ct_accumulated = ckks::add(ct_accumulated, ct_prod_diag_vec);
        }
        if (i != matrix_width - 1) {
            if (full_width) {
                ct_vec_rotating = ckks::rotate(ct_vec_rotating, rot_key_set[1]);
            } else {
                const size_t next_step = i + 1;
                ct_vec_rotating = ckks::add(ckks::rotate(ct_vec, rot_key_set[next_step]),
                                            ckks::rotate(ct_vec, rot_key_set[next_step + slot_count - matrix_width]));
            }
        }
    }
    ckks::rescale_inplace(ct_accumulated);
    return ct_accumulated;
}

#ifdef MATVEC_AMD_EXT
// "short" with the rotations written as ONE call of the batched form: the same words
static CkksCt diag_loop_batched(const std::vector<CkksPt> &diags, const CkksCt &ct_vec, const std::vector<RotKey> &rot_key_set,
                                size_t slot_count) {
    const size_t w = diags.size();
    std::vector<const RlweKsk *> keys;
    std::vector<size_t> steps;
    for (size_t s = 1; s < w; s++)
        for (size_t step : {s, s + slot_count - w}) {
            keys.push_back(&rot_key_set[step]);
            steps.push_back(rot_key_set[step].step);
        }
    std::vector<CkksCt> rot = amd::rotate(ct_vec, keys, steps);
    CkksCt acc = ckks::mult_plain(ct_vec, diags[0]);
    for (size_t i = 1; i < w; i++) acc = ckks::add(acc, ckks::mult_plain(ckks::add(rot[2 * (i - 1)], rot[2 * (i - 1) + 1]), diags[i]));
    ckks::rescale_inplace(acc);
    return acc;
}
#endif

int main(int argc, char **argv) {
    const size_t logn = argc > 1 ? std::atoi(argv[1]) : 15, L = argc > 2 ? std::atoi(argv[2]) : 10, width = argc > 3 ? std::atoi(argv[3]) : 16;
    const std::string mode = argc > 4 ? argv[4] : "short";
    const size_t reps = argc > 5 ? std::atoi(argv[5]) : 3;
    const size_t n = (size_t)1 << logn, slots = n / 2;
    const bool full = mode == "full";
    if (L < 2 || L > 10 || logn < 3 || logn > 15 || width < 2 || reps < 1 || (!full && (mode != "short" || width > slots / 2))) {
        std::fprintf(stderr, "usage: diag_matvec [3<=logN<=15] [2<=L<=10] [2<=width(<=N/4 for short)] [short|full] [reps]\n");
        return 2;
    }
    std::vector<u64> q{P50[1]};
    for (size_t k = 1; k < L; k++) q.push_back(P40[k - 1]);
    std::vector<u64> mext(q);
    mext.push_back(P50[0]);
    sm_state = 4242;
    // the rotation key set: one entry per step, filled for the steps the loop needs (mv_mul_requiring_steps, linear_algebra.h:26-35)
    std::vector<RotKey> rot_key_set(slots);
    std::vector<size_t> need;
    if (full) need.push_back(1);
    else
        for (size_t s = 1; s < width; s++) { need.push_back(s); need.push_back(s + slots - width); }
    for (size_t step : need) {
        RotKey &k = rot_key_set[step];
        for (size_t j = 0; j < L; j++) k.push_back(RlweCt{random_poly(n, mext, PolyRepForm::value), random_poly(n, mext, PolyRepForm::value)});
        k.step = step;
    }
    CkksCt ct_vec(RlweCt{random_poly(n, q, PolyRepForm::value), random_poly(n, q, PolyRepForm::value)});
    ct_vec.scaling_factor = 1099511627776.0;   // 2^40
    std::vector<CkksPt> diags;
    for (size_t i = 0; i < width; i++) {
        diags.emplace_back(random_poly(n, q, PolyRepForm::coeff));
        diags.back().scaling_factor = 1099511627776.0;
    }
#ifdef MATVEC_AMD_EXT
    // the diagonals are encoded once and used for many vectors: resident on the device (mult_plain copies and transforms its plaintext,
    // ckks/arith.cpp:47-49; a host-only diagonal would be copied on the host and uploaded in every call).  MATVEC_HOST_DIAGS=1: leave them.
    if (!std::getenv("MATVEC_HOST_DIAGS"))
        for (auto &d : diags) amd::prefetch(d);
    amd::prefetch(ct_vec[0]);
    amd::prefetch(ct_vec[1]);
#endif
    const size_t rotations = full ? width - 1 : 2 * (width - 1);
    std::printf("shape N=%zu L=%zu width=%zu mode=%s: %zu rotations under %zu keys, %zu plaintext products per product vector\n", n, L, width,
                mode.c_str(), rotations, need.size(), width);

#ifdef MATVEC_AMD_EXT
    const int passes = full ? 2 : 3;
    const char *names[3] = {"eager", "deferred", "batched-form"};
#else
    const int passes = 1;
    const char *names[1] = {"loop"};
#endif
    const char *only = std::getenv("MATVEC_PASS");   // one pass alone (a profile of it): 0 eager, 1 deferred, 2 batched-form
    for (int pass = 0; pass < passes; pass++) {
        if (only && std::atoi(only) != pass) continue;
#ifdef MATVEC_AMD_EXT
        amd::set_deferred(pass == 1);
#endif
        double best = 1e30, host = 0;
        CkksCt out;
        for (size_t r = 0; r < reps; r++) {   // (the first pass uploads the operands and the keys; they stay resident)
            device_done();
            const double t0 = now_ms();
#ifdef MATVEC_AMD_EXT
            out = pass == 2 ? diag_loop_batched(diags, ct_vec, rot_key_set, slots) : diag_loop(diags, ct_vec, rot_key_set, slots, full);
#else
            out = diag_loop(diags, ct_vec, rot_key_set, slots, full);
#endif
            const double t_host = now_ms() - t0;
            device_done();
            const double t = now_ms() - t0;
            if (r > 0 || reps == 1) {
                if (t < best) host = t_host;
                best = t < best ? t : best;
            }
        }
        std::printf("%s digest %016llx\n", names[pass], (unsigned long long)digest(out));
        std::printf("%s %.3f ms per product vector (%.3f ms per rotation; the calls returned after %.3f ms)\n", names[pass], best,
                    best / (double)rotations, host);
    }
#ifdef MATVEC_AMD_EXT
    amd::set_deferred(false);
    const auto st = amd::transfer_stats();
    std::printf("parity level %s\n", amd::parity_level_a() ? "A" : "B");
    std::printf("layer: lanes %d engine_calls %llu lane_waits %llu deferred_calls %llu deferred_groups %llu many_key_groups %llu\n", amd::lanes(),
                st.engine_calls, st.lane_waits, st.deferred_calls, st.deferred_groups, st.deferred_many_key_groups);
#endif
    return 0;
}
