// host_api_test.cpp -- the reference's own unit tests for this path, restated against the
// hehub-compatible host layer (hehub_amd/host/hehub.hpp) which runs everything on the GPU.
// Mirrors: tests/mod_arith_t.cpp:6-78, tests/ntt_t.cpp:18-181, tests/common_t.cpp:39-61,
// tests/ckks_t.cpp:136-175 (exact rescale rounding), plus raw-word comparisons of the scheme-level
// calls against the oracle (test infrastructure) and the reference's error behaviour.
// Built and run by tests/test_host_api.py (compile-only without a GPU).
#include "hehub.hpp"

#include "../../oracle/hehub_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace hehub;

static int g_fail = 0, g_checks = 0;
#define REQUIRE(c)                                                              \
    do {                                                                        \
        g_checks++;                                                             \
        if (!(c)) { g_fail++; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); } \
    } while (0)
#define REQUIRE_THROWS_AS(expr, T)                                              \
    do {                                                                        \
        g_checks++;                                                             \
        bool ok__ = false;                                                      \
        try { expr; } catch (const T &) { ok__ = true; } catch (...) {}         \
        if (!ok__) { g_fail++; std::printf("FAIL %s:%d: expected throw: %s\n", __FILE__, __LINE__, #expr); } \
    } while (0)

static u64 pow_mod(u64 q, u64 b, u64 e) {
    u64 r = 1;
    b %= q;
    while (e) {
        if (e & 1) r = (u128)r * b % q;
        b = (u128)b * b % q;
        e >>= 1;
    }
    return r;
}
static u64 bit_rev(u64 x, int bits) {
    u64 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
static u64 g_state = 88172645463325252ull;
static u64 rnd() {   // xorshift64: any deterministic stream will do
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
    return g_state;
}

static void test_batched_barrett() {   // tests/mod_arith_t.cpp:6-32
    const size_t LEN = 1024;
    for (u64 q : {65537ull, 33333333ull, 777777777777777ull, 1234567890111111111ull}) {
        std::vector<u64> v(LEN), c;
        v[0] = 42;
        for (size_t i = 1; i < LEN; i++) v[i] = v[i - 1] * 6364136223846793005ull + 1442695040888963407ull;
        c = v;
        batched_barrett_lazy(q, LEN, v.data());
        for (size_t i = 0; i < LEN; i++) { REQUIRE(v[i] < 2 * q); REQUIRE(v[i] % q == c[i] % q); }
    }
}

static void test_batched_mul_mod() {   // tests/mod_arith_t.cpp:34-59
    const size_t LEN = 1000;
    const u64 q = 1234567890111111111ull;
    std::vector<u64> f(LEN), g(LEN), h1(LEN), h2(LEN);
    f[0] = 1234; g[0] = 5678;
    for (size_t i = 1; i < LEN; i++) { f[i] = (f[i - 1] * 31 + 7) % q; g[i] = (g[i - 1] * 29 + 11) % q; }
    batched_mul_mod_hybrid(q, LEN, f.data(), g.data(), h1.data());
    batched_mul_mod_barrett(q, LEN, f.data(), g.data(), h2.data());
    for (size_t i = 0; i < LEN; i += 7) {
        REQUIRE(h1[i] == (u64)((u128)f[i] * g[i] % q));
        REQUIRE(h2[i] == (u64)((u128)f[i] * g[i] % q));
    }
}

static void test_montgomery() {   // tests/mod_arith_t.cpp:61-78
    const size_t LEN = 1000;
    const u64 q = 1099510054913ull;
    std::vector<u128> in(LEN);
    std::vector<u64> out(LEN);
    for (auto &x : in) x = ((u128)(rnd() % q) << 64) | rnd();
    batched_montgomery_128_lazy(q, LEN, in.data(), out.data());
    const u128 r = ((u128)1 << 64) % q;
    for (size_t i = 0; i < LEN; i++) { REQUIRE(out[i] < 2 * q); REQUIRE((u64)((u128)(out[i] % q) * r % q) == (u64)(in[i] % q)); }
}

static void test_ntt() {   // tests/ntt_t.cpp:18-181
    for (size_t LOGN : {4, 7, 11, 13, 14, 15}) {
        for (u64 Q : {65537ull, 260898817ull, 35184358850561ull, 36028796997599233ull, 576460752272228353ull}) {
            const size_t N = (size_t)1 << LOGN;
            if ((Q - 1) % (2 * N)) {
                std::vector<u64> z(N, 0);
                REQUIRE_THROWS_AS(ntt_negacyclic_inplace_lazy(LOGN, Q, z.data()), std::invalid_argument);
                continue;
            }
            std::vector<u64> poly(N, 0);
            poly[0] = 1;   // "one"
            ntt_negacyclic_inplace_lazy(LOGN, Q, poly.data());
            bool ok = true;
            for (auto v : poly) ok = ok && v < 2 * Q && v % Q == 1;
            REQUIRE(ok);
            std::fill(poly.begin(), poly.end(), 0);
            poly[1] = 1;   // "just x": ntt(X)[i] = psi^(2*bitrev(i)+1)
            ntt_negacyclic_inplace_lazy(LOGN, Q, poly.data());
            u64 g = 2;
            while (pow_mod(Q, g, (Q - 1) / 2) != Q - 1) g++;
            const u64 psi = pow_mod(Q, g, (Q - 1) / (2 * N));
            for (size_t i = 0; i < N; i = i * 13 + 1) REQUIRE(poly[i] % Q == pow_mod(Q, psi, 2 * bit_rev(i, (int)LOGN) + 1));
            std::vector<u64> orig(N);   // round trip on random data
            for (auto &c : orig) c = rnd() % Q;
            poly = orig;
            ntt_negacyclic_inplace_lazy(LOGN, Q, poly.data());
            intt_negacyclic_inplace_lazy(LOGN, Q, poly.data());
            ok = true;
            for (size_t i = 0; i < N; i++) { ok = ok && poly[i] < 2 * Q; poly[i] -= (poly[i] >= Q) ? Q : 0; ok = ok && poly[i] == orig[i]; }
            REQUIRE(ok);
        }
    }
    // RnsPolynomial overload + rep_form bookkeeping
    RnsPolynomial p(4096, 3, std::vector<u64>{1099510054913ull, 1099507695617ull, 1099506515969ull});
    for (auto &c : p) for (auto &w : c) w = rnd() % 1099506515969ull;
    auto q(p);
    ntt_negacyclic_inplace_lazy(q);
    REQUIRE(q.rep_form == PolyRepForm::value);
    intt_negacyclic_inplace(q);
    REQUIRE(q.rep_form == PolyRepForm::coeff);
    REQUIRE((const RnsIntVec &)q == (const RnsIntVec &)p);
    std::vector<u64> z(16, 0);
    REQUIRE_THROWS_AS(ntt_negacyclic_inplace_lazy(4, 1234567890111111111ull, z.data()), std::invalid_argument);
}

static void test_rns_polynomial() {   // tests/common_t.cpp:39-61
    RnsPolynomial r1(4096, 3, std::vector<u64>{3, 5, 7});
    RnsPolyParams params{4096, 3, std::vector<u64>{3, 5, 7}};
    RnsPolynomial r2(params);
    REQUIRE(r1.component_count() == 3 && r2.dimension() == 4096 && r2.log_dimension() == 12);
    r2.add_components({11});
    REQUIRE(r2.component_count() == 4 && r2.modulus_at(3) == 11);
    r2.remove_components(2);
    REQUIRE(r2.component_count() == 2 && r2.modulus_vec().size() == 2);
    REQUIRE_THROWS_AS(RnsPolynomial(4095, 3, std::vector<u64>{3, 5, 7}), std::invalid_argument);
    REQUIRE_THROWS_AS(RnsPolynomial(4096, 4, std::vector<u64>{3, 5, 7}), std::invalid_argument);
    RnsPolynomial a(8, 1, std::vector<u64>{65537}), b(8, 1, std::vector<u64>{65537});
    b.rep_form = PolyRepForm::value;
    REQUIRE_THROWS_AS(a += b, std::invalid_argument);
    REQUIRE_THROWS_AS(a * b, std::invalid_argument);
    RnsPolynomial c(8, 1, std::vector<u64>{12289});
    REQUIRE_THROWS_AS((RnsIntVec &)a += (const RnsIntVec &)c, std::invalid_argument);
}

// CRT-compose three residues (moduli < 2^35) into a u128
static u128 compose3(const u64 r[3], const u64 q[3]) {
    u128 x = r[0];
    u64 inv01 = pow_mod(q[1], q[0] % q[1], q[1] - 2);
    u64 t1 = (u64)((u128)((r[1] + q[1] - (u64)(x % q[1])) % q[1]) * inv01 % q[1]);
    x += (u128)q[0] * t1;
    u128 q01 = (u128)q[0] * q[1];
    u64 inv2 = pow_mod(q[2], (u64)(q01 % q[2]), q[2] - 2);
    u64 t2 = (u64)((u128)((r[2] + q[2] - (u64)(x % q[2])) % q[2]) * inv2 % q[2]);
    return x + q01 * t2;
}

static void test_ckks_rescaling() {   // tests/ckks_t.cpp:136-175
    const size_t N = 8;
    const u64 q[3] = {17179672577ull, 17179410433ull, 17176854529ull};   // create_params(8, {34,34,34})
    std::vector<u64> moduli(q, q + 3);
    ckks::CkksCt ct;
    ct.scaling_factor = std::pow(2.0, 80);
    u128 composed[2][N];
    for (int h = 0; h < 2; h++) {
        ct[h] = RnsPolynomial(N, 3, moduli);
        for (size_t k = 0; k < 3; k++) for (auto &w : ct[h][(int)k]) w = rnd() % q[k];
        for (size_t i = 0; i < N; i++) { u64 r[3] = {ct[h][0][i], ct[h][1][i], ct[h][2][i]}; composed[h][i] = compose3(r, q); }
        ntt_negacyclic_inplace_lazy(ct[h]);
    }
    ckks::rescale_inplace(ct);
    REQUIRE(ct[0].component_count() == 2 && ct[1].component_count() == 2);
    REQUIRE(std::abs(ct.scaling_factor - std::pow(2.0, 80) / q[2]) < std::pow(2.0, -60) * std::pow(2.0, 46));
    for (int h = 0; h < 2; h++) {
        intt_negacyclic_inplace_lazy(ct[h]);
        reduce_strict(ct[h]);
        for (size_t i = 0; i < N; i++) {
            u128 expect = (composed[h][i] + q[2] / 2) / q[2];
            REQUIRE(ct[h][0][i] == (u64)(expect % q[0]));
            REQUIRE(ct[h][1][i] == (u64)(expect % q[1]));
        }
    }
    REQUIRE_THROWS_AS(ckks::rescale_inplace(ct, 0), std::invalid_argument);
    bool threw_cstr = false;
    try { ckks::rescale_inplace(ct, 2); } catch (const char *) { threw_cstr = true; }
    REQUIRE(threw_cstr);
    ckks::CkksCt one;
    one[0] = RnsPolynomial(N, 1, moduli); one[1] = RnsPolynomial(N, 1, moduli);
    REQUIRE_THROWS_AS(ckks::rescale_inplace(one), std::invalid_argument);
}

static void flatten(const RnsPolynomial &p, std::vector<u64> &out) {
    for (auto &c : p) out.insert(out.end(), c.begin(), c.end());
}

static void test_scheme_level_vs_oracle() {   // raw lazy words of ckks::mult+rescale and bgv mult+mod switch
    const size_t logn = 10, N = 1 << logn, L = 3;
    std::vector<u64> mext{1099510054913ull, 1099507695617ull, 1099506515969ull, 1125899904679937ull};
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    ckks::CkksCt a, b;
    bgv::BgvCt ba, bb;
    for (int h = 0; h < 2; h++) {
        a[h] = RnsPolynomial(N, L, q); b[h] = RnsPolynomial(N, L, q);
        for (size_t k = 0; k < L; k++) { for (auto &w : a[h][(int)k]) w = rnd() % q[k]; for (auto &w : b[h][(int)k]) w = rnd() % q[k]; }
        a[h].rep_form = b[h].rep_form = PolyRepForm::value;
        ba[h] = a[h]; bb[h] = b[h];
    }
    ba.plain_modulus = bb.plain_modulus = 65537;
    RlweKsk key(L);
    for (auto &s : key) for (auto &p : s) {
        p = RnsPolynomial(N, L + 1, mext);
        for (size_t k = 0; k <= L; k++) for (auto &w : p[(int)k]) w = rnd() % mext[k];
        p.rep_form = PolyRepForm::value;
    }
    std::vector<u64> f1, f2, fk;
    for (int h = 0; h < 2; h++) { flatten(a[h], f1); flatten(b[h], f2); }
    for (auto &s : key) for (auto &p : s) flatten(p, fk);

    auto r = ckks::mult(a, b, key);
    ckks::rescale_inplace(r);
    std::vector<u64> got, exp(2 * (L - 1) * N);
    for (int h = 0; h < 2; h++) flatten(r[h], got);
    REQUIRE(orc_ckks_mult_relin_rescale(logn, L, mext.data(), f1.data(), f2.data(), fk.data(), exp.data()) == 0);
    REQUIRE(got == exp);

    auto br = bgv::relinearize(bgv::mult_low_level(ba, bb), key);
    bgv::mod_switch_inplace(br);
    got.clear();
    for (int h = 0; h < 2; h++) flatten(br[h], got);
    REQUIRE(orc_bgv_mult_relin_modswitch(logn, L, mext.data(), 65537, f1.data(), f2.data(), fk.data(), exp.data()) == 0);
    REQUIRE(got == exp);
    REQUIRE(br.plain_modulus == 65537);

    // rotation / conjugation compose the same kernels with an NTT-domain gather (ckks/arith.cpp:75-93)
    auto rot = ckks::rotate(a, key, 3);
    std::vector<u64> cyc(2 * L * N), ext(2 * (L + 1) * N), low(2 * L * N);
    orc_poly_cycle(logn, L, 3, f1.data(), cyc.data());
    orc_poly_cycle(logn, L, 3, f1.data() + L * N, cyc.data() + L * N);
    REQUIRE(orc_ext_prod_montgomery(logn, L, mext.data(), cyc.data() + L * N, fk.data(), ext.data()) == 0);
    REQUIRE(orc_ckks_rescale_by_one_prime(logn, L + 1, mext.data(), ext.data(), low.data()) == 0);
    orc_poly_add_inplace(N, L, mext.data(), low.data(), cyc.data());
    got.clear();
    for (int h = 0; h < 2; h++) flatten(rot[h], got);
    REQUIRE(got == low);
    REQUIRE(orc_ckks_rotate(logn, L, mext.data(), 3, f1.data(), fk.data(), ext.data()) == 0);
    REQUIRE(std::equal(got.begin(), got.end(), ext.begin()));
    REQUIRE(rot.scaling_factor == a.scaling_factor);
    {   // the RotKey overload carries the step with the key (keys.h:63-68, ckks.h:303)
        RotKey rk;
        for (auto &sample : key) rk.push_back(sample);
        rk.step = 3;
        auto rot2 = ckks::rotate(a, rk);
        std::vector<u64> got2;
        for (int h = 0; h < 2; h++) flatten(rot2[h], got2);
        REQUIRE(got2 == got);
    }

    auto cj = ckks::conjugate(b, key);
    REQUIRE(orc_ckks_conjugate(logn, L, mext.data(), f2.data(), fk.data(), ext.data()) == 0);
    got.clear();
    for (int h = 0; h < 2; h++) flatten(cj[h], got);
    REQUIRE(std::equal(got.begin(), got.end(), ext.begin()));
    auto coeff = a;
    coeff[1].rep_form = PolyRepForm::coeff;
    REQUIRE_THROWS_AS(ckks::rotate(coeff, key, 1), std::invalid_argument);

    RlweKsk bad(L - 1);
    REQUIRE_THROWS_AS(ext_prod_montgomery(a[0], RlweKsk()), std::invalid_argument);
    bb.plain_modulus = 257;
    REQUIRE_THROWS_AS(bgv::mult_low_level(ba, bb), std::invalid_argument);
    b.scaling_factor = 3.0;
    REQUIRE_THROWS_AS(ckks::add(a, b), std::invalid_argument);
}

static void test_plain_ops_and_decrypt_core() {   // ckks/arith.cpp:22-53, bgv/arith.cpp:17-57, rlwe.cpp:74-81, rns_transform.cpp
    const size_t logn = 9, N = 1 << logn, L = 3;
    std::vector<u64> q{1099510054913ull, 1099507695617ull, 1099506515969ull};
    ckks::CkksCt ct;
    for (int h = 0; h < 2; h++) {
        ct[h] = RnsPolynomial(N, L, q);
        for (size_t k = 0; k < L; k++) for (auto &w : ct[h][(int)k]) w = rnd() % q[k];
        ct[h].rep_form = PolyRepForm::value;
    }
    ct.scaling_factor = 1024.0;
    ckks::CkksPt pt(N, L, q);
    for (size_t k = 0; k < L; k++) for (auto &w : pt[(int)k]) w = rnd() % q[k];
    pt.rep_form = PolyRepForm::coeff;
    pt.scaling_factor = 1024.0;
    std::vector<u64> c0, c1, ptn;
    flatten(ct[0], c0); flatten(ct[1], c1); flatten(pt, ptn);
    REQUIRE(orc_poly_ntt(logn, L, q.data(), ptn.data()) == 0);

    {   // add_plain / sub_plain touch c0 only; mult_plain multiplies both and the scaling factors
        auto r = ckks::add_plain(ct, pt);
        std::vector<u64> exp(c0), got;
        orc_poly_add_inplace(N, L, q.data(), exp.data(), ptn.data());
        flatten(r[0], got);
        REQUIRE(got == exp);
        got.clear(); flatten(r[1], got);
        REQUIRE(got == c1);
        REQUIRE(r.scaling_factor == 1024.0);
        r = ckks::sub_plain(ct, pt);
        exp = c0;
        orc_poly_sub_inplace(N, L, q.data(), exp.data(), ptn.data());
        got.clear(); flatten(r[0], got);
        REQUIRE(got == exp);
        r = ckks::mult_plain(ct, pt);
        std::vector<u64> e0(L * N), e1(L * N);
        orc_poly_mul(N, L, q.data(), c0.data(), ptn.data(), e0.data());
        orc_poly_mul(N, L, q.data(), c1.data(), ptn.data(), e1.data());
        got.clear(); flatten(r[0], got);
        REQUIRE(got == e0);
        got.clear(); flatten(r[1], got);
        REQUIRE(got == e1);
        REQUIRE(r.scaling_factor == 1024.0 * 1024.0);
        ckks::CkksPt other(pt);
        other.scaling_factor = 3.0;
        REQUIRE_THROWS_AS(ckks::add_plain(ct, other), std::invalid_argument);
    }
    {   // decrypt_core = strict(INTT(c0 + c1*sk))
        RlweSk sk(N, L, q);
        for (size_t k = 0; k < L; k++) for (auto &w : sk[(int)k]) w = rnd() % q[k];
        sk.rep_form = PolyRepForm::value;
        std::vector<u64> fct(c0), fsk, exp(L * N), got;
        fct.insert(fct.end(), c1.begin(), c1.end());
        flatten(sk, fsk);
        REQUIRE(orc_rlwe_decrypt_core(logn, L, q.data(), fct.data(), fsk.data(), exp.data()) == 0);
        auto back = decrypt_core(ct, sk);
        flatten(back, got);
        REQUIRE(got == exp);
        REQUIRE(back.rep_form == PolyRepForm::coeff);
    }
    {   // BGV: plaintext modulo t lifted into the ciphertext moduli (rns_base_transform one -> many), then as above
        const u64 t = 65537;
        bgv::BgvCt bct;
        bct[0] = ct[0]; bct[1] = ct[1];
        bct.plain_modulus = t;
        bgv::BgvPt bpt(N, 1, std::vector<u64>{t});
        for (auto &w : bpt[0]) w = rnd() % t;
        bpt.rep_form = PolyRepForm::coeff;
        std::vector<u64> lifted(L * N);
        orc_rns_base_from_single(N, t, L, q.data(), bpt[0].data(), lifted.data());
        REQUIRE(orc_poly_ntt(logn, L, q.data(), lifted.data()) == 0);
        auto r = bgv::add_plain(bct, bpt);
        std::vector<u64> exp(c0), got;
        orc_poly_add_inplace(N, L, q.data(), exp.data(), lifted.data());
        flatten(r[0], got);
        REQUIRE(got == exp);
        REQUIRE(r.plain_modulus == t);
        r = bgv::mult_plain(bct, bpt);
        std::vector<u64> e1(L * N);
        orc_poly_mul(N, L, q.data(), c1.data(), lifted.data(), e1.data());
        got.clear(); flatten(r[1], got);
        REQUIRE(got == e1);
        bgv::BgvPt wrong(N, 1, std::vector<u64>{257});
        wrong.rep_form = PolyRepForm::coeff;
        REQUIRE_THROWS_AS(bgv::add_plain(bct, wrong), std::invalid_argument);

        // many -> one: small coefficients come back modulo t; NTT-form input is a logic_error (rns_transform.cpp:108-111)
        RnsPolynomial small(N, L, q);
        for (size_t i = 0; i < N; i++) {
            const long long v = (long long)(rnd() % 2001) - 1000;
            for (size_t k = 0; k < L; k++) small[(int)k][i] = v >= 0 ? (u64)v : q[k] - (u64)(-v);
        }
        small.rep_form = PolyRepForm::coeff;
        std::vector<u64> fin, exp1(N);
        flatten(small, fin);
        REQUIRE(orc_rns_base_to_single_small(N, L, q.data(), t, fin.data(), exp1.data()) == 1);
        auto one = rns_base_transform(small, std::vector<u64>{t});
        REQUIRE(std::equal(exp1.begin(), exp1.end(), one[0].begin()));
        // coefficients that are not small: the CRT branch (big integers in the reference, mixed radix on the device)
        RnsPolynomial big(N, L, q);
        for (size_t k = 0; k < L; k++) for (auto &w : big[(int)k]) w = rnd() % q[k];
        big.rep_form = PolyRepForm::coeff;
        std::vector<u64> fbig, expb(N);
        flatten(big, fbig);
        orc_rns_base_to_single(N, L, q.data(), t, fbig.data(), expb.data());
        auto oneb = rns_base_transform(big, std::vector<u64>{t});
        REQUIRE(std::equal(expb.begin(), expb.end(), oneb[0].begin()));
        small.rep_form = PolyRepForm::value;
        REQUIRE_THROWS_AS(rns_base_transform(small, std::vector<u64>{t}), std::logic_error);
    }
}

static void test_ragged_operands() {   // rns.cpp:59-72 (+= on the first self.component_count() limbs of b), :120-140 (min components)
    const size_t N = 256;
    std::vector<u64> q{1099510054913ull, 1099507695617ull, 1099506515969ull};
    RnsPolynomial a(N, 2, q), b(N, 3, q);
    for (size_t k = 0; k < 2; k++) for (auto &w : a[(int)k]) w = rnd() % (2 * q[k]);
    for (size_t k = 0; k < 3; k++) for (auto &w : b[(int)k]) w = rnd() % (2 * q[k]);
    a.rep_form = b.rep_form = PolyRepForm::value;
    std::vector<u64> fa, fb;
    flatten(a, fa); flatten(b, fb);
    {   // a += b uses b's first two limbs
        auto s = a;
        s += b;
        std::vector<u64> exp(fa), got;
        orc_poly_add_inplace(N, 2, q.data(), exp.data(), fb.data());
        flatten(s, got);
        REQUIRE(s.component_count() == 2 && got == exp);
    }
    {   // b += a is an error: a has fewer components than b
        auto s = b;
        REQUIRE_THROWS_AS(s += a, std::invalid_argument);
    }
    {   // a * b and b * a both have two components
        auto p1 = a * b, p2 = b * a;
        std::vector<u64> exp(2 * N), got;
        orc_poly_mul(N, 2, q.data(), fa.data(), fb.data(), exp.data());
        flatten(p1, got);
        REQUIRE(p1.component_count() == 2 && got == exp);
        got.clear(); flatten(p2, got);
        REQUIRE(p2.component_count() == 2 && got == exp);
    }
    {   // empty polynomial: operators are no-ops, not errors
        RnsPolynomial e1(N, 0, std::vector<u64>{}), e2(N, 0, std::vector<u64>{});
        e1.rep_form = e2.rep_form = PolyRepForm::value;
        e1 += e2;
        auto p = e1 * e2;
        REQUIRE(p.component_count() == 0);
    }
}


// Device residency of the mirror's vectors (hehub.hpp; allocator.h:105-220 is what hehub does on the host): value semantics
// and host access must behave exactly as if the words had never left the host, while PCIe is crossed only when somebody looks.
// amd::set_parity_level_a: the scheme-level calls of the object API return canonical residues = reduce_strict of the level-B words
static void test_parity_level_a() {
    const size_t N = 4096;
    const std::vector<u64> q{1125899903827969ull, 1099510054913ull, 1099507695617ull};   // a 50-bit (wide) and two 40-bit moduli
    auto fill = [&](RnsPolynomial &p, u64 seed) {
        for (size_t k = 0; k < p.component_count(); k++)
            for (size_t i = 0; i < N; i++) p[(int)k][i] = (seed * 1000003ull + k * 7919ull + i * 104729ull) % q[k];
        p.rep_form = PolyRepForm::value;
    };
    ckks::CkksCt ct;
    for (int h = 0; h < 2; h++) { ct[h] = RnsPolynomial(N, 3, q); fill(ct[h], 11 + h); }
    ct.scaling_factor = std::pow(2.0, 80);
    REQUIRE(!amd::parity_level_a());
    ckks::CkksCt b = ct;
    ckks::rescale_inplace(b);
    amd::set_parity_level_a(true);
    REQUIRE(amd::parity_level_a());
    ckks::CkksCt a = ct;
    ckks::rescale_inplace(a);
    amd::set_parity_level_a(false);
    bool same = true;
    for (int h = 0; h < 2; h++)
        for (size_t k = 0; k < 2; k++)
            for (size_t i = 0; i < N; i++) {
                const u64 wb = b[h].view((int)k)[i], wa = a[h].view((int)k)[i];
                same = same && wa == (wb >= q[k] ? wb - q[k] : wb) && wa < q[k];
            }
    REQUIRE(same);   // (a Harvey product lands in [q, 2q) only rarely, so most words coincide; tests/test_gpu_level_a.py has inputs where they do not)
}

static void test_device_residency() {
    const size_t N = 4096;
    const std::vector<u64> q{1099510054913ull, 1099507695617ull, 1099506515969ull};
    auto fill = [&](RnsPolynomial &p, u64 seed) {
        for (size_t k = 0; k < p.component_count(); k++)
            for (size_t i = 0; i < N; i++) p[(int)k][i] = (seed * 1000003ull + k * 7919ull + i * 104729ull) % q[k];
    };
    RnsPolynomial a(N, 3, q), b(N, 3, q);
    fill(a, 1); fill(b, 2);
    const RnsPolynomial a_host(a);          // host-only copies to compare against
    REQUIRE(!a.device_resident() && !a_host.device_resident());
    auto s0 = amd::transfer_stats();
    ntt_negacyclic_inplace_lazy(a);         // uploads a once, result stays in HBM
    ntt_negacyclic_inplace_lazy(b);
    auto s1 = amd::transfer_stats();
    REQUIRE(a.device_resident() && b.device_resident());
    REQUIRE(s1.h2d_bytes - s0.h2d_bytes == 2 * 3 * N * 8 && s1.d2h_bytes == s0.d2h_bytes);
    RnsPolynomial c = a * b;                // operands and result on the device: nothing crosses
    c += a;
    RnsPolynomial d(c);                     // deep copy, device to device
    d *= (u64)3;                            // must not touch c
    auto s2 = amd::transfer_stats();
    REQUIRE(s2.h2d_bytes == s1.h2d_bytes && s2.d2h_bytes == s1.d2h_bytes && c.device_resident() && d.device_resident());
    // the same computation on words that go through the host after every step
    RnsPolynomial ah(a_host), bh(N, 3, q);
    fill(bh, 2);
    ntt_negacyclic_inplace_lazy(ah); (void)ah[0][0];
    ntt_negacyclic_inplace_lazy(bh); (void)bh[0][0];
    RnsPolynomial ch = ah * bh; (void)ch[0][0];
    ch += ah; (void)ch[0][0];
    RnsPolynomial dh(ch);
    dh *= (u64)3;
    REQUIRE((const RnsIntVec &)c == (const RnsIntVec &)ch);      // (fetches c: one download)
    REQUIRE((const RnsIntVec &)d == (const RnsIntVec &)dh);
    REQUIRE(!((const RnsIntVec &)c == (const RnsIntVec &)d));
    auto s3 = amd::transfer_stats();
    // a const look keeps both copies current: the next engine call uploads nothing
    const RnsPolynomial &cc = c;
    const u64 w = cc[1][17];
    c += a;
    auto s4 = amd::transfer_stats();
    REQUIRE(s4.h2d_bytes == s3.h2d_bytes);
    // view(): the same read-only look on a NON-const vector -- no invalidation, no upload at the next call
    const u64 w2 = c.view(1)[17] + c.view()[0][0];
    (void)w2;
    c += a;
    auto s4b = amd::transfer_stats();
    REQUIRE(s4b.h2d_bytes == s3.h2d_bytes && s4b.device_copies_invalidated == s4.device_copies_invalidated);
    // ... while operator[] on a non-const vector has to assume a write: counted, and the next call uploads the vector again
    { RnsPolynomial g(c); (void)g.view(0)[0]; g += a; auto t0 = amd::transfer_stats(); const u64 r = g[0][0]; (void)r; g += a;
      auto t1 = amd::transfer_stats();
      REQUIRE(t1.device_copies_invalidated == t0.device_copies_invalidated + 1 && t1.h2d_bytes - t0.h2d_bytes == 3 * N * 8); }
    // a writable look makes the host copy the current one: the next engine call uploads it, and sees the change
    RnsPolynomial e(c);
    e[1][17] = (w + 1) % q[1];
    auto s5 = amd::transfer_stats();
    e += a;
    auto s6 = amd::transfer_stats();
    REQUIRE(s6.h2d_bytes - s5.h2d_bytes == 3 * N * 8);
    RnsPolynomial f(c);
    f += a;
    REQUIRE(!((const RnsIntVec &)e == (const RnsIntVec &)f));   // e differs from f in exactly the word that was written
    {
        const RnsPolynomial &ce = e, &cf = f;
        size_t diff = 0;
        for (int k = 0; k < 3; k++)
            for (size_t i = 0; i < N; i++) diff += ce[k][i] != cf[k][i];
        REQUIRE(diff == 1);
    }
    // move leaves the source empty (allocator.h:137-155); the target keeps the device words
    RnsPolynomial g(std::move(f));
    REQUIRE(f.component_count() == 0 && f.dimension() == 0 && g.component_count() == 3 && g.device_resident() == false);
    RnsPolynomial h = a * b;
    RnsPolynomial h2(std::move(h));
    REQUIRE(h.component_count() == 0 && h2.device_resident());
    // remove_components / add_components on words that are in HBM only
    RnsPolynomial r = a * b;
    const RnsPolynomial r_full(r);
    r.remove_components();
    REQUIRE(r.component_count() == 2 && r.modulus_vec().size() == 2 && r.device_resident());
    {
        const RnsPolynomial &cr = r, &cf = r_full;
        bool same = true;
        for (int k = 0; k < 2; k++) same = same && cr[k] == cf[k];
        REQUIRE(same);
    }
    RnsPolynomial r2 = a * b;
    r2.add_components({65537});
    REQUIRE(r2.component_count() == 4 && r2.modulus_at(3) == 65537 && !r2.device_resident());
    {
        const RnsPolynomial &cr = r2, &cf = r_full;
        bool same = true, zero = true;
        for (int k = 0; k < 3; k++) same = same && cr[k] == cf[k];
        for (size_t i = 0; i < N; i++) zero = zero && cr[3][i] == 0;
        REQUIRE(same && zero);
    }
    // assignment over a device-resident object, self-assignment
    RnsPolynomial t1 = a * b, t2(N, 3, q);
    t2 = t1;
    t1 *= (u64)5;
    REQUIRE((const RnsIntVec &)t2 == (const RnsIntVec &)r_full);
    t2 = *&t2;
    REQUIRE((const RnsIntVec &)t2 == (const RnsIntVec &)r_full);
    // the halves of a ciphertext that an engine call produced feed the next scheme-level call without a copy: a chain of
    // additions and a product crosses PCIe for the inputs only
    RlweCt x{a, b}, y{b, a};
    auto s7 = amd::transfer_stats();
    RlweCt z = add(x, y);
    for (int i = 0; i < 5; i++) z = add(z, x);
    z = sub(z, y);
    auto s8 = amd::transfer_stats();
    REQUIRE(s8.h2d_bytes == s7.h2d_bytes && s8.d2h_bytes == s7.d2h_bytes && z[0].device_resident() && z[1].device_resident());
    RnsPolynomial chk0 = a;   // the same operations one polynomial at a time
    chk0 += b;
    for (int i = 0; i < 5; i++) chk0 += a;
    chk0 -= b;
    REQUIRE((const RnsIntVec &)z[0] == (const RnsIntVec &)chk0);
}


// ---- hehub_amd_ext.hpp: batched forms and lanes ----------------------------------------------------------------------------
static bool same_words(const RlweCt &x, const RlweCt &y) {
    std::vector<u64> fx, fy;
    for (int h = 0; h < 2; h++) { flatten(x[h], fx); flatten(y[h], fy); }
    return x[0].modulus_vec() == y[0].modulus_vec() && x[0].rep_form == y[0].rep_form && fx == fy;
}
struct BatchFixture {
    size_t logn = 11, N = 1 << 11, L = 3, B = 5;
    std::vector<u64> mext{1099510054913ull, 1099507695617ull, 1099506515969ull, 1125899904679937ull}, q;
    std::vector<ckks::CkksCt> a, b;
    RlweKsk key;
    BatchFixture() : q(mext.begin(), mext.begin() + 3), key(3) {
        for (size_t i = 0; i < B; i++) {
            ckks::CkksCt x, y;
            for (int h = 0; h < 2; h++) {
                x[h] = RnsPolynomial(N, L, q); y[h] = RnsPolynomial(N, L, q);
                for (size_t k = 0; k < L; k++) { for (auto &w : x[h][(int)k]) w = rnd() % q[k]; for (auto &w : y[h][(int)k]) w = rnd() % q[k]; }
                x[h].rep_form = y[h].rep_form = PolyRepForm::value;
            }
            x.scaling_factor = std::pow(2.0, 40) * (double)(i + 1);
            y.scaling_factor = std::pow(2.0, 40);
            a.push_back(std::move(x)); b.push_back(std::move(y));
        }
        for (auto &smp : key) for (auto &p : smp) {
            p = RnsPolynomial(N, L + 1, mext);
            for (size_t k = 0; k <= L; k++) for (auto &w : p[(int)k]) w = rnd() % mext[k];
            p.rep_form = PolyRepForm::value;
        }
    }
};

// every batched form returns, element by element, the words of the single call of hehub's interface (which the tests above pin
// to the oracle), whatever the lane count; operands that a batched call gathered or produced are found packed by the next one
static void test_batched_forms() {
    BatchFixture f;
    const size_t B = f.B;
    std::vector<ckks::CkksCt> single_m, single_mr, single_rot, single_cj, single_add, single_sub;
    for (size_t i = 0; i < B; i++) {
        single_m.push_back(ckks::mult(f.a[i], f.b[i], f.key));
        single_mr.push_back(single_m.back());
        ckks::rescale_inplace(single_mr.back());
        single_rot.push_back(ckks::rotate(f.a[i], f.key, 5));
        single_cj.push_back(ckks::conjugate(f.b[i], f.key));
    }
    const auto before = amd::transfer_stats();
    auto m = amd::mult(f.a, f.b, f.key);
    auto mr = amd::mult_rescale(f.a, f.b, f.key);
    auto rot = amd::rotate(f.a, f.key, 5);
    auto cj = amd::conjugate(f.b, f.key);
    REQUIRE(m.size() == B && mr.size() == B && rot.size() == B && cj.size() == B);
    for (size_t i = 0; i < B; i++) {
        REQUIRE(same_words(m[i], single_m[i]) && m[i].scaling_factor == single_m[i].scaling_factor);
        REQUIRE(same_words(mr[i], single_mr[i]) && mr[i].scaling_factor == single_mr[i].scaling_factor);
        REQUIRE(same_words(rot[i], single_rot[i]) && rot[i].scaling_factor == f.a[i].scaling_factor);
        REQUIRE(same_words(cj[i], single_cj[i]));
    }
    // results of a batched call feed the next one: add / sub / rescale on the products (equal scaling factors pairwise)
    for (size_t i = 0; i < B; i++) {
        single_add.push_back(ckks::add(single_m[i], single_m[i]));
        single_sub.push_back(ckks::sub(single_m[i], single_m[i]));
        ckks::rescale_inplace(single_add.back());
    }
    auto sum = amd::add(m, m), dif = amd::sub(m, m);
    amd::rescale_inplace(sum);
    for (size_t i = 0; i < B; i++) {
        REQUIRE(same_words(sum[i], single_add[i]) && sum[i].scaling_factor == single_add[i].scaling_factor);
        REQUIRE(same_words(dif[i], single_sub[i]));
        REQUIRE(sum[i][0].component_count() == f.L - 1);
    }
    // nothing crossed PCIe for all of that except the looks at the words (the operands were on the device already)
    REQUIRE(amd::transfer_stats().h2d_bytes == before.h2d_bytes);

    // bgv: mult_low_level + relinearize [+ mod_switch_inplace]
    std::vector<bgv::BgvCt> ba, bb, s1, s2;
    for (size_t i = 0; i < B; i++) {
        bgv::BgvCt x(RlweCt{f.a[i][0], f.a[i][1]}), y(RlweCt{f.b[i][0], f.b[i][1]});
        x.plain_modulus = y.plain_modulus = 65537;
        ba.push_back(std::move(x)); bb.push_back(std::move(y));
        s1.push_back(bgv::relinearize(bgv::mult_low_level(ba.back(), bb.back()), f.key));
        s2.push_back(s1.back());
        bgv::mod_switch_inplace(s2.back());
    }
    auto g1 = amd::mult(ba, bb, f.key), g2 = amd::mult_mod_switch(ba, bb, f.key);
    auto g3 = g1;
    amd::mod_switch_inplace(g3);
    for (size_t i = 0; i < B; i++) {
        REQUIRE(same_words(g1[i], s1[i]) && g1[i].plain_modulus == 65537);
        REQUIRE(same_words(g2[i], s2[i]) && g2[i].plain_modulus == 65537);
        REQUIRE(same_words(g3[i], s2[i]));
    }

    // a batch without one common shape is the loop of single calls; the checks of the single call are kept
    auto ragged = f.a;
    ragged[2][0].remove_components();
    ragged[2][1].remove_components();
    auto rr = amd::rotate(std::vector<ckks::CkksCt>{f.a[0], f.a[1]}, f.key, 1);
    REQUIRE(same_words(rr[1], ckks::rotate(f.a[1], f.key, 1)));
    REQUIRE_THROWS_AS(amd::rotate(ragged, f.key, 1), std::invalid_argument);          // member 2 has no key for its level
    auto coeff = f.a;
    coeff[3][1].rep_form = PolyRepForm::coeff;
    REQUIRE_THROWS_AS(amd::mult(coeff, f.b, f.key), std::invalid_argument);           // "Operand a is in coefficient form."
    REQUIRE_THROWS_AS(amd::rotate(coeff, f.key, 1), std::invalid_argument);
    REQUIRE_THROWS_AS(amd::add(f.a, f.b), std::invalid_argument);                     // scaling factors mismatch (member 1 on)
    REQUIRE_THROWS_AS(amd::mult(f.a, std::vector<ckks::CkksCt>(f.b.begin(), f.b.begin() + 2), f.key), std::invalid_argument);
    bb[1].plain_modulus = 257;
    REQUIRE_THROWS_AS(amd::mult(ba, bb, f.key), std::invalid_argument);               // "Plain moduli mismatch."
    std::vector<ckks::CkksCt> one_limb;
    for (size_t i = 0; i < 2; i++) {
        ckks::CkksCt x = f.a[i];
        x[0].remove_components(2); x[1].remove_components(2);
        one_limb.push_back(std::move(x));
    }
    REQUIRE_THROWS_AS(amd::rescale_inplace(one_limb), std::invalid_argument);         // "Unable to drop the only one prime."
    REQUIRE(amd::mult(std::vector<ckks::CkksCt>(), std::vector<ckks::CkksCt>(), f.key).empty());
}

// independent chains of single calls over 1 and over several lanes: same words, and the lanes are really used
static void test_lanes() {
    BatchFixture f;
    const int before = amd::lanes();
    std::vector<std::vector<ckks::CkksCt>> res;
    for (int lanes : {1, 3, 8}) {
        amd::set_lanes(lanes);
        REQUIRE(amd::lanes() == lanes);
        std::vector<ckks::CkksCt> x(f.a.begin(), f.a.end());
        ckks::CkksCt acc;
        for (int it = 0; it < 3; it++)
            for (size_t c = 0; c < x.size(); c++) {
                x[c] = ckks::rotate(ckks::mult(x[c], f.b[c], f.key), f.key, c + 1);
                x[c][0] *= (u64)(it + 2);             // an in-place operator on one half (its own lane's block)
            }
        // a fan-in: one call reads what several lanes produced
        acc = ckks::mult(x[0], x[1], f.key);
        for (size_t c = 2; c < x.size(); c++) acc = ckks::mult(acc, x[c], f.key);
        x.push_back(acc);
        // and a batched call on what the lanes produced
        auto rot = amd::rotate(x, f.key, 2);
        x.insert(x.end(), rot.begin(), rot.end());
        res.push_back(std::move(x));
    }
    amd::synchronize();
    for (size_t v = 1; v < res.size(); v++)
        for (size_t i = 0; i < res[0].size(); i++) REQUIRE(same_words(res[0][i], res[v][i]));
    REQUIRE(amd::transfer_stats().lane_waits > 0);   // the fan-in needed them
    amd::set_lanes(before);
}

// device ranks (hehub_amd_ext.hpp: set_devices; here the ranks share the GPU): the program of test_lanes with the calls spread over 1 / 2 /
// 3 ranks -- independent chains land on different ranks, the fan-in reads what several ranks produced (operands move over), a batched
// call is cut into per-rank slices -- returns the words of one rank, call by call and recorded; a rank whose device does not exist fails
// the calls that are routed there with hehub-style exceptions and leaves the layer usable
static void test_devices() {
    const int before = amd::devices();
    if (before != 1) return;   // (the process was started with HEHUB_AMD_DEVICES: its ranks keep their devices; this test sets its own)
    BatchFixture f;
    const bool was = amd::deferred();
    std::vector<std::vector<ckks::CkksCt>> res;
    for (int mode = 0; mode < 2; mode++)
        for (int ranks : {1, 2, 3}) {
            amd::set_deferred(mode == 1);
            amd::set_devices(std::vector<int>((size_t)ranks, 0));
            REQUIRE(amd::devices() == ranks);
            const auto s0 = amd::transfer_stats();
            // (fresh copies of the inputs made on the HOST side of the fixture: every configuration uploads its own operands -- an operand
            // that is already resident somewhere draws the calls that read it to its rank)
            auto host_copies = [](const std::vector<ckks::CkksCt> &src) {
                std::vector<ckks::CkksCt> v;
                for (auto &ct : src) {
                    ckks::CkksCt c(ct);
                    for (int h = 0; h < 2; h++) (void)c[h][0][0];   // a writable look: the copy's words are host words again
                    v.push_back(std::move(c));
                }
                return v;
            };
            std::vector<ckks::CkksCt> x = host_copies(f.a), b = host_copies(f.b);
            for (int it = 0; it < 2; it++)
                for (size_t c = 0; c < x.size(); c++) x[c] = ckks::rotate(ckks::mult(x[c], b[c], f.key), f.key, c + 1);
            ckks::CkksCt acc = ckks::mult(x[0], x[1], f.key);
            for (size_t c = 2; c < x.size(); c++) acc = ckks::mult(acc, x[c], f.key);
            x.push_back(acc);
            auto rot = amd::rotate(x, f.key, 2);          // a batch: contiguous slices, one per rank
            x.insert(x.end(), rot.begin(), rot.end());
            auto sq = amd::mult_rescale(x, x, f.key);
            x.insert(x.end(), sq.begin(), sq.end());
            amd::synchronize();
            const auto s1 = amd::transfer_stats();
            for (int r = 0; r < ranks; r++) REQUIRE(s1.calls_by_device[r] > s0.calls_by_device[r]);   // every rank worked
            if (ranks > 1) REQUIRE(s1.peer_copies > s0.peer_copies);                                     // the fan-in moved operands
            res.push_back(std::move(x));
        }
    for (size_t v = 1; v < res.size(); v++) {
        REQUIRE(res[v].size() == res[0].size());
        for (size_t i = 0; i < res[0].size(); i++) REQUIRE(same_words(res[0][i], res[v][i]));
    }
    res.clear();
    // a device that does not exist: the call routed to that rank throws, the others work, and the layer recovers
    REQUIRE_THROWS_AS(amd::set_devices(std::vector<int>{}), std::invalid_argument);
    REQUIRE_THROWS_AS(amd::set_devices(std::vector<int>(9, 0)), std::invalid_argument);
    amd::set_deferred(false);
    amd::set_devices(std::vector<int>{0, 0, 0, 4242});
    int thrown = 0, fine = 0;
    for (int i = 0; i < 8; i++) {
        ckks::CkksCt c(f.a[0]);
        for (int h = 0; h < 2; h++) (void)c[h][0][0];
        try {
            auto r = ckks::rotate(c, f.key, 1);
            (void)r[0].view(0)[0];
            fine++;
        } catch (const std::runtime_error &) {
            thrown++;
        }
    }
    REQUIRE(thrown == 2);   // round robin over four ranks: every fourth host-only call went to the rank without a device
    REQUIRE(fine == 6);
    REQUIRE_THROWS_AS(amd::set_devices(2), std::logic_error);   // (rank 1 was made on HIP device 0: asking for device 1 there now is refused)
    REQUIRE(amd::devices() == 4);
    amd::set_devices(std::vector<int>{0, 0});
    ckks::CkksCt again = ckks::rotate(f.a[0], f.key, 1);
    amd::set_devices(1);
    ckks::CkksCt again2 = ckks::rotate(f.a[0], f.key, 1);
    REQUIRE(same_words(again, again2));
    amd::set_deferred(was);
}

// deferred mode: the same calls recorded and run as batches -- the words, the argument checks and the object state of the eager calls
static void test_deferred() {
    BatchFixture f;
    const size_t B = f.B;
    const bool was = amd::deferred();
    // eager reference: a loop of independent mults + rescale, interleaved chains, an accumulate chain
    amd::set_deferred(false);
    std::vector<ckks::CkksCt> e_out, e_chain(f.a.begin(), f.a.end());
    for (size_t i = 0; i < B; i++) {
        e_out.push_back(ckks::mult(f.a[i], f.b[i], f.key));
        ckks::rescale_inplace(e_out.back());
    }
    for (int it = 0; it < 2; it++)
        for (size_t c = 0; c < B; c++) e_chain[c] = ckks::rotate(ckks::mult(e_chain[c], f.b[c], f.key), f.key, 1);
    ckks::CkksCt e_sum = ckks::mult(f.a[0], f.b[0], f.key);
    for (size_t i = 1; i < B; i++) {
        auto p = ckks::mult(f.a[i], f.b[i], f.key);
        p.scaling_factor = e_sum.scaling_factor;
        e_sum = ckks::add(e_sum, p);
    }
    ckks::CkksCt e_cj = ckks::sub(ckks::conjugate(f.a[1], f.key), f.a[1]);
    auto mixed_chain = [&] {   // sums and differences in one chain
        ckks::CkksCt x = f.b[0];
        x = ckks::add(x, f.b[1]);
        x = ckks::sub(x, f.b[2]);
        x = ckks::add(x, f.b[3]);
        return x;
    };
    ckks::CkksCt e_mix = mixed_chain();
    amd::synchronize();

    amd::set_deferred(true);
    REQUIRE(amd::deferred());
    const auto st0 = amd::transfer_stats();
    std::vector<ckks::CkksCt> d_out, d_chain(f.a.begin(), f.a.end());
    for (size_t i = 0; i < B; i++) {
        d_out.push_back(ckks::mult(f.a[i], f.b[i], f.key));
        ckks::rescale_inplace(d_out.back());
        // object state is there at once, words are not asked for
        REQUIRE(d_out.back()[0].component_count() == f.L - 1 && d_out.back()[1].modulus_vec().size() == f.L - 1);
        REQUIRE(d_out.back().scaling_factor == e_out[i].scaling_factor);
    }
    REQUIRE(amd::transfer_stats().deferred_calls == st0.deferred_calls);   // nothing has run yet
    for (int it = 0; it < 2; it++)
        for (size_t c = 0; c < B; c++) d_chain[c] = ckks::rotate(ckks::mult(d_chain[c], f.b[c], f.key), f.key, 1);
    ckks::CkksCt d_sum = ckks::mult(f.a[0], f.b[0], f.key);
    for (size_t i = 1; i < B; i++) {
        auto p = ckks::mult(f.a[i], f.b[i], f.key);
        p.scaling_factor = d_sum.scaling_factor;
        d_sum = ckks::add(d_sum, p);
    }
    ckks::CkksCt d_cj = ckks::sub(ckks::conjugate(f.a[1], f.key), f.a[1]);
    ckks::CkksCt d_mix = mixed_chain();
    // plaintext products (ckks::mult_plain = two operator* calls, ckks/arith.cpp:47-53) are recorded too: the diagonal loop of
    // src/circuits/linear_algebra.h:109-133 (rotate, mult_plain, add) never runs the queue by itself
    ckks::CkksPt d_pt;
    {
        RnsPolynomial ptp(f.N, f.L, f.q);
        for (size_t k = 0; k < f.L; k++) for (auto &w : ptp[(int)k]) w = rnd() % f.q[k];
        ptp.rep_form = PolyRepForm::coeff;
        d_pt = ckks::CkksPt(std::move(ptp));
        d_pt.scaling_factor = std::pow(2.0, 40);
    }
    std::vector<ckks::CkksCt> d_lin;
    for (size_t i = 0; i < B; i++) d_lin.push_back(ckks::mult_plain(ckks::rotate(f.a[i], f.key, i + 1), d_pt));
    REQUIRE(amd::transfer_stats().deferred_calls == st0.deferred_calls);   // still nothing has run
    // a copy of a result that has not been computed yet is recorded too (examples/ckks_example.cpp: `ct_sum = ct_prod`)
    ckks::CkksCt d_copy = d_sum, d_copy2;
    d_copy2 = d_out[0];
    REQUIRE(amd::transfer_stats().deferred_calls == st0.deferred_calls);   // still nothing has run
    // the argument checks of the single calls are made when the call is recorded
    auto coeff = f.a[0];
    coeff[1].rep_form = PolyRepForm::coeff;
    REQUIRE_THROWS_AS(ckks::mult(coeff, f.b[0], f.key), std::invalid_argument);
    REQUIRE_THROWS_AS(ckks::rotate(coeff, f.key, 1), std::invalid_argument);
    REQUIRE_THROWS_AS(ckks::relinearize(ckks::mult_low_level(f.a[0], f.b[0]), RlweKsk()), std::invalid_argument);
    auto other_scale = f.a[1];
    other_scale.scaling_factor = 3.0;
    REQUIRE_THROWS_AS(ckks::add(f.a[1], other_scale), std::invalid_argument);
    // a look at one word runs the queue; the recorded calls ran as a few batched engine calls
    REQUIRE(same_words(d_out[B - 1], e_out[B - 1]));
    const auto st1 = amd::transfer_stats();
    REQUIRE(st1.deferred_calls - st0.deferred_calls >= 3 * B + 2 * 3 * B);
    REQUIRE(st1.deferred_groups - st0.deferred_groups < (st1.deferred_calls - st0.deferred_calls) / 2);
    REQUIRE(st1.deferred_fused - st0.deferred_fused >= B);   // the mult + rescale_inplace loop ran as the engine's one-call pipeline
    for (size_t i = 0; i < B; i++) {
        REQUIRE(same_words(d_out[i], e_out[i]));
        REQUIRE(same_words(d_chain[i], e_chain[i]));
    }
    REQUIRE(same_words(d_sum, e_sum) && same_words(d_cj, e_cj) && same_words(d_mix, e_mix));
    // the accumulate chain (`sum = add(sum, p)`: src/circuits/linear_algebra.h:117-121) and the mixed one each ran as ONE pass over their terms
    REQUIRE(st1.deferred_chain_sums - st0.deferred_chain_sums == (B - 1) + 3);
    REQUIRE(same_words(d_copy, e_sum) && same_words(d_copy2, e_out[0]) && d_copy2.scaling_factor == e_out[0].scaling_factor);
    amd::set_deferred(false);
    for (size_t i = 0; i < B; i++) {
        auto e = ckks::mult_plain(ckks::rotate(f.a[i], f.key, i + 1), d_pt);
        REQUIRE(same_words(d_lin[i], e) && d_lin[i].scaling_factor == e.scaling_factor);
    }
    amd::set_deferred(true);
    // an eager in-place operator on an operand of a recorded call: the recorded call saw the words as they were
    ckks::CkksCt x = f.a[2];
    auto prod = ckks::mult(x, f.b[2], f.key);
    x[0] *= (u64)3;
    REQUIRE(same_words(prod, ckks::mult(f.a[2], f.b[2], f.key)));
    // ... while an eager call on a vector no recorded call knows about leaves the queue alone
    {
        auto pend = ckks::mult(f.a[3], f.b[3], f.key);
        const auto q0 = amd::transfer_stats();
        RnsPolynomial other(f.N, f.L, f.q);
        for (size_t k = 0; k < f.L; k++) for (auto &w : other[(int)k]) w = rnd() % f.q[k];
        other *= (u64)5;
        REQUIRE(amd::transfer_stats().deferred_calls == q0.deferred_calls);   // nothing ran
        REQUIRE(same_words(pend, ckks::mult(f.a[3], f.b[3], f.key)));
        REQUIRE(amd::transfer_stats().deferred_calls > q0.deferred_calls);
        // the in-place transforms of polynomials are recorded too (the plaintext NTTs inside mult_plain run as one batch): several of
        // them, and an eager operator on a recorded result -- the words of the eager calls
        std::vector<RnsPolynomial> rec(3, other), eag(3, other);
        const auto q1 = amd::transfer_stats();
        for (auto &p : rec) ntt_negacyclic_inplace_lazy(p);
        REQUIRE(amd::transfer_stats().deferred_calls == q1.deferred_calls);   // recorded, not run
        rec[1] *= (u64)7;                                                     // (an eager write to a pending result runs the queue first)
        intt_negacyclic_inplace(rec[2]);
        amd::set_deferred(false);
        for (auto &p : eag) ntt_negacyclic_inplace_lazy(p);
        eag[1] *= (u64)7;
        intt_negacyclic_inplace(eag[2]);
        amd::set_deferred(true);
        for (size_t i = 0; i < 3; i++) REQUIRE((rec[i] == eag[i] && rec[i].rep_form == eag[i].rep_form));
        REQUIRE(amd::transfer_stats().deferred_calls >= q1.deferred_calls + 4);
    }
    // what the ENGINE refuses (a modulus the transforms reject: 2N does not divide q - 1) is refused by the RECORDING call since round 6
    // (hp_check_chain at record time; round 5 let the call through and failed when the queue ran): std::invalid_argument at the call, as
    // hehub (ntt.cpp:26-29) and the call-by-call mode throw it; the ciphertext is untouched, what was recorded before still runs
    {
        const std::vector<u64> badq{1099511627689ull, 1099511627563ull};   // 40-bit primes, not = 1 mod 2N
        ckks::CkksCt bad;
        for (int h = 0; h < 2; h++) {
            bad[h] = RnsPolynomial(f.N, 2, badq);
            for (size_t k = 0; k < 2; k++) for (auto &w : bad[h][(int)k]) w = rnd() % badq[k];
            bad[h].rep_form = PolyRepForm::value;
        }
        auto good = ckks::mult(f.a[4], f.b[4], f.key);         // recorded
        ckks::CkksCt victim = bad;
        const auto q0 = amd::transfer_stats();
        REQUIRE_THROWS_AS(ckks::rescale_inplace(victim), std::invalid_argument);
        REQUIRE(amd::transfer_stats().deferred_calls == q0.deferred_calls);   // (the refusal ran nothing)
        REQUIRE(victim[0].component_count() == 2);
        REQUIRE(same_words(victim, bad));
        auto again = ckks::mult(f.a[4], f.b[4], f.key);
        REQUIRE(same_words(again, good));
        amd::set_deferred(false);
        REQUIRE(same_words(again, ckks::mult(f.a[4], f.b[4], f.key)));
        REQUIRE_THROWS_AS(ckks::rescale_inplace(bad), std::invalid_argument);   // call by call: the same refusal, at the call
        amd::set_deferred(true);
    }
    // bgv: mult_low_level + relinearize + mod_switch_inplace
    bgv::BgvCt ba(RlweCt{f.a[0][0], f.a[0][1]}), bb(RlweCt{f.b[0][0], f.b[0][1]});
    ba.plain_modulus = bb.plain_modulus = 65537;
    auto db = bgv::relinearize(bgv::mult_low_level(ba, bb), f.key);
    bgv::mod_switch_inplace(db);
    amd::set_deferred(false);   // runs what is pending
    auto eb = bgv::relinearize(bgv::mult_low_level(ba, bb), f.key);
    bgv::mod_switch_inplace(eb);
    REQUIRE(same_words(db, eb) && db.plain_modulus == 65537);
    // what the ENGINE refuses is refused when the call is recorded, not when the queue runs (round 6: hp_check_chain at record time): a
    // modulus the transforms cannot use (ntt.cpp:26-29: 2N does not divide q - 1) throws std::invalid_argument at the call in both modes,
    // nothing is recorded, the object is unchanged and the queue keeps working
    for (int mode = 0; mode < 2; mode++) {
        amd::set_deferred(mode == 1);
        const std::vector<u64> bad{1099510054913ull, 1234567890111111111ull};   // the second: 60.1 bits, 2N does not divide q - 1
        RnsPolynomial p(f.N, 2, bad);
        for (size_t k = 0; k < 2; k++)
            for (auto &w : p[(int)k]) w = rnd() % bad[k];
        const RnsPolynomial before(p);
        const auto s0 = amd::transfer_stats();
        REQUIRE_THROWS_AS(ntt_negacyclic_inplace_lazy(p), std::invalid_argument);
        REQUIRE(p.rep_form == PolyRepForm::coeff);
        REQUIRE(p == before);
        ckks::CkksCt badct(RlweCt{p, p});
        for (int h = 0; h < 2; h++) badct[h].rep_form = PolyRepForm::value;
        REQUIRE_THROWS_AS(ckks::rescale_inplace(badct), std::invalid_argument);
        REQUIRE(badct[0].component_count() == 2);                       // (the drop did not happen)
        const std::vector<u64> even{1099510054913ull, 1099510054912ull};   // Montgomery products need an odd modulus
        RnsPolynomial e1(f.N, 2, even), e2(f.N, 2, even);
        e1.rep_form = e2.rep_form = PolyRepForm::value;
        REQUIRE_THROWS_AS((void)(e1 * e2), std::invalid_argument);
        REQUIRE(amd::transfer_stats().deferred_calls == s0.deferred_calls);
        auto fine = ckks::rotate(f.a[0], f.key, 1);                     // ... and the layer goes on
        (void)fine[0].view(0)[0];
    }
    amd::set_deferred(false);
    // bgv plain operations (bgv/arith.cpp:17-57: the plaintext is lifted into the ciphertext's moduli, transformed and combined): the
    // lift (rns_base_transform, one modulus -> many) and += / -= of polynomials are recorded too -- a loop of them never runs the queue
    {
        std::vector<bgv::BgvPt> pts;
        for (size_t i = 0; i < 4; i++) {
            bgv::BgvPt p(f.N, 1, std::vector<u64>{65537});
            for (auto &w : p[0]) w = rnd() % 65537;
            p.rep_form = PolyRepForm::coeff;
            pts.push_back(std::move(p));
        }
        for (auto &p : pts) amd::prefetch(p);
        auto plain_loop = [&](std::vector<bgv::BgvCt> &out) {
            for (size_t i = 0; i < 4; i++) {
                out.push_back(bgv::add_plain(ba, pts[i]));
                out.push_back(bgv::sub_plain(bb, pts[i]));
                out.push_back(bgv::mult_plain(out[3 * i], pts[i]));   // (reads a recorded sum)
            }
        };
        std::vector<bgv::BgvCt> e, d;
        plain_loop(e);
        RnsPolynomial es = f.a[4][0];
        es += f.b[4][0];
        es -= f.a[4][1];
        es += es;
        amd::synchronize();
        amd::set_deferred(true);
        const auto q0 = amd::transfer_stats();
        plain_loop(d);
        RnsPolynomial ds = f.a[4][0];
        ds += f.b[4][0];
        ds -= f.a[4][1];
        ds += ds;   // (both operands the same recorded sum)
        REQUIRE(amd::transfer_stats().deferred_calls == q0.deferred_calls);   // recorded, nothing ran
        REQUIRE_THROWS_AS(bgv::add_plain(ba, bgv::BgvPt(f.N, 1, std::vector<u64>{257})), std::invalid_argument);   // checked at the call
        RnsPolynomial shorter(f.N, f.L - 1, f.q);
        REQUIRE_THROWS_AS(ds += shorter, std::invalid_argument);
        REQUIRE(ds == es);
        const auto q1 = amd::transfer_stats();
        // 12 lifts as one batch, 12 transforms as one, the sums / differences / products in a few groups
        REQUIRE(q1.deferred_calls - q0.deferred_calls >= 12 + 12 + 8 + 8 + 3);
        REQUIRE(q1.deferred_groups - q0.deferred_groups <= 16 * (unsigned long long)amd::devices());   // (a group belongs to one device rank)
        for (size_t i = 0; i < d.size(); i++) REQUIRE((same_words(d[i], e[i]) && d[i].plain_modulus == 65537));
        amd::set_deferred(false);
    }
    // a chain of sums and differences longer than one launch's 32 terms (examples/ckks_example.cpp accumulates 10 000)
    {
        auto long_chain = [&] {
            ckks::CkksCt x = f.b[0];
            for (size_t i = 0; i < 40; i++) x = (i % 3 == 2) ? ckks::sub(x, f.b[i % B]) : ckks::add(x, f.b[(i + 1) % B]);
            return x;
        };
        amd::set_deferred(false);
        ckks::CkksCt e_long = long_chain();
        amd::synchronize();
        amd::set_deferred(true);
        const auto q0 = amd::transfer_stats();
        ckks::CkksCt d_long = long_chain();
        REQUIRE(amd::transfer_stats().deferred_calls == q0.deferred_calls);
        REQUIRE(same_words(d_long, e_long));
        REQUIRE(amd::transfer_stats().deferred_chain_sums - q0.deferred_chain_sums == 40);
        amd::set_deferred(false);
    }
    amd::set_deferred(was);
}

int main() {
    struct { const char *name; void (*fn)(); } tests[] = {
        {"batched_barrett", test_batched_barrett}, {"batched_mul_mod", test_batched_mul_mod}, {"montgomery", test_montgomery},
        {"ntt", test_ntt}, {"rns_polynomial", test_rns_polynomial}, {"ckks_rescaling", test_ckks_rescaling},
        {"scheme_level_vs_oracle", test_scheme_level_vs_oracle}, {"plain_ops_and_decrypt_core", test_plain_ops_and_decrypt_core},
        {"ragged_operands", test_ragged_operands}, {"device_residency", test_device_residency}, {"parity_level_a", test_parity_level_a},
        {"batched_forms", test_batched_forms}, {"lanes", test_lanes}, {"devices", test_devices}, {"deferred", test_deferred}};
    for (auto &t : tests) {
        try {
            t.fn();
        } catch (const std::exception &e) {
            g_fail++;
            std::printf("FAIL test %s: unexpected exception: %s\n", t.name, e.what());
        } catch (...) {
            g_fail++;
            std::printf("FAIL test %s: unexpected exception\n", t.name);
        }
    }
    std::printf("%s: %d checks, %d failures\n", g_fail ? "FAILED" : "All tests passed", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
