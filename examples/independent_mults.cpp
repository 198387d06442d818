// independent_mults.cpp -- what a caller of hehub's one-ciphertext-per-call interface gets from the GPU when its operations are
// INDEPENDENT (the loops of src/circuits/linear_algebra.h:109-133, bench/benchmarks.cpp:24-35, examples/ckks_example.cpp:15-26),
// three ways, same words each way:
//
//   serial   for i: out[i] = ckks::mult(a[i], b[i], relin_key); ckks::rescale_inplace(out[i]);       hehub's API as it is
//   batch    out = amd::mult_rescale(a, b, relin_key);                                               hehub_amd_ext.hpp: ONE engine call
//   chains   C independent chains  x[c] = ckks::rotate(ckks::mult(x[c], b[c], relin_key), relin_key, 1)  interleaved call by call,
//            with the layer's lanes (amd::set_lanes) at 1 and at `lanes`: how much independent single calls overlap on the device
//
// Ciphertext and key words are synthetic (splitmix64): the program checks RING ARITHMETIC, word for word; it prints an FNV-1a-64
// digest over every word of the results, which must be the same for every mode AND for hehub itself on the CPU
// (make -C oracle ref_indep -> oracle/_ref/ref_indep_cpu: this file against hehub's own headers, serial and chains modes).
//
//   independent_mults [logN=15] [L=10] [B=256] [mode=all|serial|batch|chains] [reps=3] [lanes=8] [chains=8] [chain_len=6]
#ifdef CHAIN_REFERENCE_HEADERS
#include "fhe/bgv/bgv.h"
#include "fhe/ckks/ckks.h"
#include "fhe/primitives/keys.h"
#ifdef INDEP_AMD_EXT
#include "hehub_amd_ext.hpp"
#endif
#else
#include "hehub.hpp"
#define INDEP_AMD_EXT 1
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    for (size_t k = 0; k < moduli.size(); k++) {
        auto &limb = p[(int)k];
        for (size_t i = 0; i < n; i++) limb[i] = splitmix() % moduli[k];
    }
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
static u64 digest(const std::vector<CkksCt> &cts) {
    u64 h = 0xcbf29ce484222325ull;
    for (const CkksCt &ct : cts) h = fnv(fnv(h, ct[0]), ct[1]);
    return h;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void device_done() {
#ifdef INDEP_AMD_EXT
    amd::synchronize();
#endif
}

int main(int argc, char **argv) {
    const size_t logn = argc > 1 ? std::atoi(argv[1]) : 15, L = argc > 2 ? std::atoi(argv[2]) : 10, B = argc > 3 ? std::atoi(argv[3]) : 256;
    const std::string mode = argc > 4 ? argv[4] : "all";
    const size_t reps = argc > 5 ? std::atoi(argv[5]) : 3, lanes = argc > 6 ? std::atoi(argv[6]) : 8;
    const size_t chains = argc > 7 ? std::atoi(argv[7]) : 8, chain_len = argc > 8 ? std::atoi(argv[8]) : 6;
    if (L < 2 || L > 10 || logn < 1 || logn > 15 || B < 1 || reps < 1) {
        std::fprintf(stderr, "usage: independent_mults [logN<=15] [2<=L<=10] [B] [all|serial|batch|chains] [reps] [lanes] [chains] [chain_len]\n");
        return 2;
    }
    const size_t n = (size_t)1 << logn;
    std::vector<u64> q{P50[1]};
    for (size_t k = 1; k < L; k++) q.push_back(P40[k - 1]);
    std::vector<u64> mext(q);
    mext.push_back(P50[0]);
    sm_state = 77;
    RlweKsk relin_key;
    for (size_t j = 0; j < L; j++) relin_key.push_back(RlweCt{random_poly(n, mext), random_poly(n, mext)});
    std::vector<CkksCt> a, b;
    for (size_t i = 0; i < B; i++) {
        a.emplace_back(RlweCt{random_poly(n, q), random_poly(n, q)});
        b.emplace_back(RlweCt{random_poly(n, q), random_poly(n, q)});
        a.back().scaling_factor = b.back().scaling_factor = 1099511627776.0;   // 2^40
    }
    std::printf("shape N=%zu L=%zu B=%zu\n", n, L, B);

    // ---- serial: hehub's interface as it is --------------------------------------------------------------------------------
    if (mode == "all" || mode == "serial") {
        std::vector<CkksCt> out;
        double best = 1e30, host = 0;
        for (size_t r = 0; r < reps; r++) {   // (the first pass uploads the operands and the key; they stay resident)
            out.clear();
            const double t0 = now_ms();
            for (size_t i = 0; i < B; i++) {
                out.push_back(ckks::mult(a[i], b[i], relin_key));
                ckks::rescale_inplace(out.back());
            }
            const double t_host = now_ms() - t0;   // the calls have returned (enqueued); the device may still be working
            device_done();
            const double t = now_ms() - t0;
            if (r > 0 || reps == 1) {
                if (t < best) host = t_host;
                best = t < best ? t : best;
            }
        }
        std::printf("serial digest %016llx\n", (unsigned long long)digest(out));
        std::printf("serial %.3f ms per hom-mult (%.0f hom-mult/s); the calls themselves returned after %.3f ms per hom-mult\n", best / B,
                    1e3 * B / best, host / B);
        // rotate, add, rescale, one ciphertext per call (the batched chain below must print the same digest)
        std::vector<CkksCt> sum1;
        for (size_t i = 0; i < B; i++) {
            CkksCt s = ckks::add(ckks::rotate(a[i], relin_key, 1), a[i]);
            ckks::rescale_inplace(s);
            sum1.push_back(std::move(s));
        }
        std::printf("serial-chain digest %016llx\n", (unsigned long long)digest(sum1));
    }
#ifdef INDEP_AMD_EXT
    // ---- batch: one engine call -----------------------------------------------------------------------------------------------
    if (mode == "all" || mode == "batch") {
        std::vector<CkksCt> out;
        double best = 1e30, first = 0;
        for (size_t r = 0; r < reps; r++) {
            const double t0 = now_ms();
            out = amd::mult_rescale(a, b, relin_key);
            device_done();
            const double t = now_ms() - t0;
            if (r == 0) first = t;
            if (r > 0 || reps == 1) best = t < best ? t : best;
        }
        std::printf("batch digest %016llx\n", (unsigned long long)digest(out));
        std::printf("batch %.3f ms per hom-mult (%.0f hom-mult/s); first call (gathers the operands into one block) %.3f ms per hom-mult\n",
                    best / B, 1e3 * B / best, first / B);
        // a batched chain: rotate, add, rescale -- the operands lie packed in one block since the first batched call gathered them
        const double t0 = now_ms();
        std::vector<CkksCt> rot = amd::rotate(a, relin_key, 1);
        std::vector<CkksCt> sum = amd::add(rot, a);
        amd::rescale_inplace(sum);
        device_done();
        const double t = now_ms() - t0;
        std::printf("batch-chain digest %016llx\n", (unsigned long long)digest(sum));
        std::printf("batch-chain rotate+add+rescale %.3f ms per ciphertext\n", t / B);
    }
#endif
    // ---- chains: independent single calls, one lane against several ------------------------------------------------------------
    if (mode == "all" || mode == "chains") {
        const size_t C = chains < B ? chains : B;
#ifdef INDEP_AMD_EXT
        const int passes = 2, lane_counts[2] = {1, (int)lanes};
#else
        const int passes = 1;
#endif
        double ms[2] = {0, 0};
        u64 dg[2] = {0, 0};
        for (int pass = 0; pass < passes; pass++) {
#ifdef INDEP_AMD_EXT
            amd::set_lanes(lane_counts[pass]);
#endif
            double best = 1e30, host = 0;
            std::vector<CkksCt> x;
            for (size_t r = 0; r < reps; r++) {
                x.assign(a.begin(), a.begin() + C);
                device_done();
                const double t0 = now_ms();
                for (size_t it = 0; it < chain_len; it++)
                    for (size_t c = 0; c < C; c++) x[c] = ckks::rotate(ckks::mult(x[c], b[c], relin_key), relin_key, 1);
                const double t_host = now_ms() - t0;
                device_done();
                const double t = now_ms() - t0;
                if (r > 0 || reps == 1) {
                    if (t < best) host = t_host;
                    best = t < best ? t : best;
                }
            }
            ms[pass] = best / (double)(C * chain_len);
            std::printf("chains pass %d: calls returned after %.3f ms per step, device done after %.3f\n", pass, host / (double)(C * chain_len), ms[pass]);
            dg[pass] = digest(x);
        }
        std::printf("chains digest %016llx\n", (unsigned long long)dg[0]);
#ifdef INDEP_AMD_EXT
        std::printf("chains-lanes digest %016llx\n", (unsigned long long)dg[1]);
        std::printf("chains %zu x %zu mult+rotate: %.3f ms per step on 1 lane, %.3f ms on %d lanes (x %.2f)\n", C, chain_len, ms[0], ms[1],
                    amd::lanes(), ms[0] / ms[1]);
        const auto st = amd::transfer_stats();
        std::printf("lane_waits %llu engine_calls %llu\n", st.lane_waits, st.engine_calls);
#else
        std::printf("chains %zu x %zu mult+rotate: %.3f ms per step\n", C, chain_len, ms[0]);
#endif
    }
#ifdef INDEP_AMD_EXT
    {
        const auto st = amd::transfer_stats();
        if (st.deferred_calls)
            std::printf("deferred: %llu recorded calls ran as %llu batched engine calls (%llu mult + rescale_inplace triples as the one-call pipeline)\n",
                        st.deferred_calls, st.deferred_groups, st.deferred_fused);
#ifndef CHAIN_REFERENCE_HEADERS
        std::printf("devices %d engine calls per device rank:", amd::devices());   // (HEHUB_AMD_DEVICES: the layer spreads independent ciphertexts over them)
        for (int r = 0; r < amd::devices(); r++) std::printf(" %llu", st.calls_by_device[r]);
        std::printf("; copies between ranks %llu (%.1f MiB)\n", st.peer_copies, st.peer_bytes / 1048576.0);
#endif
    }
#endif
    return 0;
}
