// random_program.cpp -- a seeded RANDOM PROGRAM over hehub's one-ciphertext-per-call interface: a pool of ciphertexts, a few hundred
// calls drawn from everything the hot path offers (ckks.h:270-313, ckks/arith.cpp:15-93, rns.h:129-151) with the aliasing, copying,
// moving and looking-at-words a real caller does in between.  The program depends on its seed only, never on HOW the layer runs it:
// one lane or eight (amd::set_lanes / HEHUB_AMD_LANES), eager or deferred (HEHUB_AMD_DEFER=1), either parity level.  It prints one
// FNV-1a-64 digest over every word it looked at and every word of the final pool; every way of running it -- and hehub itself on the
// CPU (make -C oracle ref_randprog -> oracle/_ref/ref_randprog_cpu: this file against hehub's own headers) -- must print the same.
//
// Ciphertext, plaintext and key words are synthetic (splitmix64): this checks ring arithmetic and the layer's bookkeeping (ready
// tickets, the block pool, placeholders of recorded calls), not cryptography.
//
//   random_program [logN=12] [L=4] [pool=12] [ops=300] [seed=1] [bgv=0]
#ifdef CHAIN_REFERENCE_HEADERS
#include "fhe/bgv/bgv.h"
#include "fhe/ckks/ckks.h"
#include "fhe/common/ntt.h"
#include "fhe/primitives/keys.h"
#else
#include "hehub.hpp"
#endif

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
// the program's own choices come from a second stream, so that the DATA of a fresh ciphertext does not depend on how many choices were made
static u64 pick_state;
static u64 pick() {
    u64 z = (pick_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static size_t pick(size_t n) { return (size_t)(pick() % n); }

// first primes of hehub's 50-bit and 40-bit rows (primelists.cpp:131, :85-86)
static const u64 P50[] = {1125899904679937ull, 1125899903827969ull};
static const u64 P40[] = {1099510054913ull, 1099507695617ull, 1099506515969ull, 1099504549889ull, 1099503894529ull,
                          1099503370241ull, 1099502714881ull, 1099502518273ull, 1099501731841ull, 1099500814337ull};

static RnsPolynomial random_poly(size_t n, const std::vector<u64> &moduli, PolyRepForm form = PolyRepForm::value) {
    RnsPolynomial p(n, moduli.size(), moduli);
    for (size_t k = 0; k < moduli.size(); k++) {
        auto &limb = p[(int)k];
        for (size_t i = 0; i < n; i++) limb[i] = splitmix() % moduli[k];
    }
    p.rep_form = form;
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

static const double SC = 1099511627776.0;   // 2^40: every result is put back on this scale (the factor is bookkeeping, not words)

template <class Ct, class Quad, bool BGV> struct Program {
    size_t n, L;
    std::vector<u64> q, mext;
    RlweKsk key;
    std::vector<Ct> pool;
    std::vector<size_t> level;   // limbs of pool[i]
    Quad quad;
    bool quad_full = false;
    u64 h = 0xcbf29ce484222325ull;
    unsigned long long count[18] = {0};

    std::vector<u64> moduli_at(size_t lv) const { return std::vector<u64>(q.begin(), q.begin() + lv); }
    void settle(Ct &ct) {
        if constexpr (BGV) ct.plain_modulus = 65537;
        else ct.scaling_factor = SC;
    }
    Ct fresh() {
        Ct ct(RlweCt{random_poly(n, q), random_poly(n, q)});
        settle(ct);
        return ct;
    }
    // a slot at level lv other than `not_this` (pool.size() = none)
    size_t slot_at(size_t lv, size_t not_this = (size_t)-1) {
        std::vector<size_t> c;
        for (size_t i = 0; i < pool.size(); i++)
            if (level[i] == lv && i != not_this) c.push_back(i);
        return c.empty() ? pool.size() : c[pick(c.size())];
    }
    void put(size_t z, Ct &&ct, size_t lv) {
        settle(ct);
        pool[z] = std::move(ct);
        level[z] = lv;
    }
    void look(size_t x) { h = fnv(fnv(h, pool[x][0]), pool[x][1]); }

    void run(size_t ops) {
        const size_t P = pool.size();
        for (size_t it = 0; it < ops; it++) {
            const size_t op = pick(18), x = pick(P), z = pick(P);
            count[op]++;
            switch (op) {
            case 0: case 1: {   // add / sub of two ciphertexts of one level; the result may replace an operand
                const size_t y = slot_at(level[x]);
                if constexpr (BGV) put(z, op == 0 ? bgv::add(pool[x], pool[y]) : bgv::sub(pool[x], pool[y]), level[x]);
                else put(z, op == 0 ? ckks::add(pool[x], pool[y]) : ckks::sub(pool[x], pool[y]), level[x]);
                break;
            }
            case 2: case 3: {   // mult [+ the drop of the last prime] at the key's level
                const size_t a = slot_at(L), b = slot_at(L);
                if (a == P) break;
                if constexpr (BGV) {
                    bgv::BgvCt r = bgv::relinearize(bgv::mult_low_level(pool[a], pool[b]), key);
                    if (op == 3) bgv::mod_switch_inplace(r);
                    put(z, std::move(r), op == 3 ? L - 1 : L);
                } else {
                    ckks::CkksCt r = ckks::mult(pool[a], pool[b], key);
                    if (op == 3) ckks::rescale_inplace(r);
                    put(z, std::move(r), op == 3 ? L - 1 : L);
                }
                break;
            }
            case 4: {   // the tensor product now, its relinearisation some calls later
                const size_t a = slot_at(L), b = slot_at(L);
                if (a == P || quad_full) break;
                if constexpr (BGV) quad = bgv::mult_low_level(pool[a], pool[b]);
                else quad = ckks::mult_low_level(pool[a], pool[b]);
                quad_full = true;
                break;
            }
            case 5: {
                if (!quad_full) break;
                if constexpr (BGV) put(z, bgv::relinearize(quad, key), L);
                else put(z, ckks::relinearize(quad, key), L);
                quad_full = false;
                break;
            }
            case 6: case 7: {   // rotations and the conjugation (the same synthetic key serves: ring arithmetic only)
                if constexpr (BGV) break;
                else {
                    const size_t a = slot_at(L);
                    if (a == P) break;
                    static const size_t steps[3] = {1, 2, 5};
                    put(z, op == 6 ? ckks::rotate(pool[a], key, steps[pick(3)]) : ckks::conjugate(pool[a], key), L);
                }
                break;
            }
            case 8: {   // drop the last prime in place
                if (level[x] < 2) break;
                if constexpr (BGV) bgv::mod_switch_inplace(pool[x]);
                else ckks::rescale_inplace(pool[x]);
                settle(pool[x]);
                level[x]--;
                break;
            }
            case 9: {   // copy
                if (z == x) break;
                pool[z] = pool[x];
                level[z] = level[x];
                break;
            }
            case 10: {   // move out of a slot, a fresh ciphertext into it
                if (z == x) break;
                pool[z] = std::move(pool[x]);
                level[z] = level[x];
                pool[x] = fresh();
                level[x] = L;
                break;
            }
            case 11: {   // plaintext operations (ckks/arith.cpp:22-53: the plaintext is transformed inside; bgv/arith.cpp:17-57: lifted first)
                if constexpr (BGV) {
                    bgv::BgvPt pt(random_poly(n, std::vector<u64>{65537}, PolyRepForm::coeff));
                    const size_t which = pick(3);
                    put(z, which == 0 ? bgv::add_plain(pool[x], pt) : which == 1 ? bgv::sub_plain(pool[x], pt) : bgv::mult_plain(pool[x], pt),
                        level[x]);
                } else {
                    ckks::CkksPt pt(random_poly(n, moduli_at(level[x]), PolyRepForm::coeff));
                    pt.scaling_factor = SC;
                    const size_t which = pick(3);
                    put(z, which == 0 ? ckks::add_plain(pool[x], pt) : which == 1 ? ckks::sub_plain(pool[x], pt) : ckks::mult_plain(pool[x], pt),
                        level[x]);
                }
                break;
            }
            case 12: look(x); break;   // somebody reads words (deferred mode: the queue runs)
            case 13: {   // operators on the polynomials themselves (rns.h:129-151)
                const size_t y = slot_at(level[x], x);
                if (y == P) break;
                const size_t which = pick(3);
                if (which == 0) pool[x][0] += pool[y][0];
                else if (which == 1) pool[x][1] -= pool[y][1];
                else {
                    const size_t half = pick(2);
                    const u64 scalar = pick() % 1000003 + 2;
                    pool[x][half] *= scalar;
                }
                break;
            }
            case 14: {   // a fresh ciphertext (keeps the top level populated)
                pool[x] = fresh();
                level[x] = L;
                break;
            }
            case 15: {   // the caller writes a word of a polynomial in place (operator[] on a non-const limb), then goes on computing with it
                const size_t half = pick(2), k = pick(level[x]), i = pick(n);
                const u64 w = pick() % q[0] % 1000;
                auto &limb = pool[x][half][(int)k];
                limb[i] = w;
                break;
            }
            case 16: {   // transforms in place on one polynomial (ntt.h:41-51, :72-92): to coefficients and back -- other lazy words, same residues
                auto &poly = pool[x][pick(2)];
                intt_negacyclic_inplace_lazy(poly);
                ntt_negacyclic_inplace_lazy(poly);
                break;
            }
            default: {   // a chain on one slot: mult, rotate (or a second mult), all at the key's level
                const size_t a = slot_at(L), b = slot_at(L);
                if (a == P) break;
                if constexpr (BGV) {
                    bgv::BgvCt r = bgv::relinearize(bgv::mult_low_level(pool[a], pool[b]), key);
                    settle(r);
                    put(z, bgv::relinearize(bgv::mult_low_level(r, pool[a]), key), L);
                } else {
                    ckks::CkksCt r = ckks::mult(pool[a], pool[b], key);
                    settle(r);
                    put(z, ckks::rotate(r, key, 1), L);
                }
                break;
            }
            }
        }
        for (size_t i = 0; i < P; i++) look(i);
    }
};

template <class Ct, class Quad, bool BGV> static int run_program(size_t logn, size_t L, size_t pool, size_t ops, u64 seed) {
    Program<Ct, Quad, BGV> p;
    p.n = (size_t)1 << logn;
    p.L = L;
    p.q.push_back(BGV ? P40[9] : P50[1]);
    for (size_t k = 1; k < L; k++) p.q.push_back(P40[k - 1]);
    p.mext = p.q;
    p.mext.push_back(P50[0]);
    sm_state = seed * 1000003ull + 17;
    pick_state = seed * 7919ull + 5;
    for (size_t j = 0; j < L; j++) p.key.push_back(RlweCt{random_poly(p.n, p.mext), random_poly(p.n, p.mext)});
    for (size_t i = 0; i < pool; i++) {
        p.pool.push_back(p.fresh());
        p.level.push_back(L);
    }
    p.run(ops);
    std::printf("program digest %016llx\n", (unsigned long long)p.h);
    std::printf("calls:");
    for (int i = 0; i < 18; i++) std::printf(" %llu", p.count[i]);
    std::printf("\n");
#ifndef CHAIN_REFERENCE_HEADERS
    const auto st = amd::transfer_stats();
    std::printf("layer: lanes %d deferred %d engine_calls %llu lane_waits %llu deferred_calls %llu deferred_groups %llu deferred_fused %llu\n", amd::lanes(),
                (int)amd::deferred(), st.engine_calls, st.lane_waits, st.deferred_calls, st.deferred_groups, st.deferred_fused);
    std::printf("devices %d engine calls per device rank:", amd::devices());
    for (int r = 0; r < amd::devices(); r++) std::printf(" %llu", st.calls_by_device[r]);
    std::printf("; copies between ranks %llu (%.1f MiB)\n", st.peer_copies, st.peer_bytes / 1048576.0);
#endif
    return 0;
}

int main(int argc, char **argv) {
    const size_t logn = argc > 1 ? std::atoi(argv[1]) : 12, L = argc > 2 ? std::atoi(argv[2]) : 4, pool = argc > 3 ? std::atoi(argv[3]) : 12;
    const size_t ops = argc > 4 ? std::atoi(argv[4]) : 300;
    const u64 seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 1;
    const bool bgv = argc > 6 && std::atoi(argv[6]) != 0;
    if (L < 2 || L > 10 || logn < 3 || logn > 15 || pool < 2) {
        std::fprintf(stderr, "usage: random_program [3<=logN<=15] [2<=L<=10] [pool>=2] [ops] [seed] [bgv]\n");
        return 2;
    }
    std::printf("shape N=%zu L=%zu pool=%zu ops=%zu seed=%llu %s\n", (size_t)1 << logn, L, pool, ops, (unsigned long long)seed, bgv ? "bgv" : "ckks");
    try {
        return bgv ? run_program<bgv::BgvCt, bgv::BgvQuadraticCt, true>(logn, L, pool, ops, seed)
                   : run_program<ckks::CkksCt, ckks::CkksQuadraticCt, false>(logn, L, pool, ops, seed);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
    } catch (const char *e) {
        std::fprintf(stderr, "exception: %s\n", e);
    }
    return 1;
}
