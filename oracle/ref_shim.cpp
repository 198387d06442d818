/*
 * ref_shim.cpp -- C-ABI glue around the UNMODIFIED primihub/hehub reference.
 *
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with the
 * reference's own sources (read in place from /root/reference, never copied)
 * into oracle/_ref/libhehub_ref.so.  It exposes the reference's hot-path
 * functions with the same flat-array signatures as hehub_oracle.h (prefix
 * ref_ instead of orc_) so tests can diff the oracle against the real thing,
 * and small timing loops used as bench.py's cpu_baseline ("kind":"reference").
 * No arithmetic lives here: every function marshals into the reference's own
 * containers and calls the reference.
 */
#include "fhe/bgv/bgv.h"
#include "fhe/ckks/ckks.h"
#include "fhe/common/mod_arith.h"
#include "fhe/common/ntt.h"
#include "fhe/common/permutation.h"
#include "fhe/common/rns.h"
#include "fhe/primitives/keys.h"
#include "fhe/primitives/rgsw.h"
#include "fhe/primitives/rlwe.h"
#include "fhe/common/rns_transform.h"

#include <chrono>
#include <cstring>
#include <vector>

using namespace hehub;

namespace {

std::vector<u64> mods(const u64 *m, size_t n) { return std::vector<u64>(m, m + n); }

RnsPolynomial load_poly(size_t n, size_t L, const u64 *moduli, const u64 *flat,
                        PolyRepForm form) {
    RnsPolynomial p(n, L, mods(moduli, L));
    for (size_t k = 0; k < L; k++) std::memcpy(p[k].data(), flat + k * n, n * sizeof(u64));
    p.rep_form = form;
    return p;
}

void store_poly(const RnsPolynomial &p, u64 *flat) {
    const size_t n = p.dimension();
    for (size_t k = 0; k < p.component_count(); k++)
        std::memcpy(flat + k * n, p[k].data(), n * sizeof(u64));
}

RlweKsk load_key(size_t n, size_t L, const u64 *moduli_ext, const u64 *key) {
    RlweKsk ksk(L);
    const size_t Le = L + 1;
    for (size_t j = 0; j < L; j++)
        for (size_t h = 0; h < 2; h++)
            ksk[j][h] = load_poly(n, Le, moduli_ext, key + ((j * 2 + h) * Le) * n, PolyRepForm::value);
    return ksk;
}

template <typename F> int guarded(F &&f) {
    try {
        f();
        return 0;
    } catch (const std::invalid_argument &) {
        return -1;
    } catch (...) {
        return -9;
    }
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

namespace hehub {
u64 __pow_mod(u64 modulus, u64 base, size_t index);
u64 __get_2nth_unity_root(u64 modulus, u64 n);
} // namespace hehub

extern "C" {

u64 ref_mul_mod_harvey_lazy(u64 q, u64 a, u64 b, u64 bh) { return mul_mod_harvey_lazy(q, a, b, bh); }
u64 ref_inverse_mod_prime(u64 elem, u64 prime) { return inverse_mod_prime(elem, prime); }
u64 ref_pow_mod(u64 q, u64 base, u64 index) { return __pow_mod(q, base, index); }
int ref_get_2nth_unity_root(u64 q, u64 n, u64 *root) {
    return guarded([&] { *root = __get_2nth_unity_root(q, n); });
}
u64 ref_bit_rev(u64 x, int bits) { return __bit_rev_naive_16(x, bits); }

void ref_batched_barrett_lazy(u64 q, size_t n, u64 *v) { batched_barrett_lazy(q, n, v); }
void ref_batched_barrett(u64 q, size_t n, u64 *v) { batched_barrett(q, n, v); }
void ref_batched_reduce_strict(u64 q, size_t n, u64 *v) { batched_reduce_strict(q, n, v); }
void ref_batched_mul_mod_hybrid_lazy(u64 q, size_t n, const u64 *a, const u64 *b, u64 *out) {
    batched_mul_mod_hybrid_lazy(q, n, a, b, out);
}
void ref_batched_mul_mod_barrett_lazy(u64 q, size_t n, const u64 *a, const u64 *b, u64 *out) {
    batched_mul_mod_barrett_lazy(q, n, a, b, out);
}
void ref_batched_montgomery_128_lazy(u64 q, size_t n, const u64 *in128, u64 *out) {
    batched_montgomery_128_lazy(q, n, reinterpret_cast<const u128 *>(in128), out);
}

int ref_ntt_negacyclic_inplace_lazy(size_t logn, u64 q, u64 *x) {
    return guarded([&] { ntt_negacyclic_inplace_lazy(logn, q, x); });
}
int ref_intt_negacyclic_inplace_lazy(size_t logn, u64 q, u64 *x) {
    return guarded([&] { intt_negacyclic_inplace_lazy(logn, q, x); });
}

void ref_poly_add_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, const u64 *b) {
    auto s = load_poly(n, L, moduli, self, PolyRepForm::value);
    s += load_poly(n, L, moduli, b, PolyRepForm::value);
    store_poly(s, self);
}
void ref_poly_sub_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, const u64 *b) {
    auto s = load_poly(n, L, moduli, self, PolyRepForm::value);
    s -= load_poly(n, L, moduli, b, PolyRepForm::value);
    store_poly(s, self);
}
void ref_poly_mul(size_t n, size_t L, const u64 *moduli, const u64 *a, const u64 *b, u64 *out) {
    auto r = load_poly(n, L, moduli, a, PolyRepForm::value) * load_poly(n, L, moduli, b, PolyRepForm::value);
    store_poly(r, out);
}
void ref_poly_scalar_mul_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, u64 scalar) {
    auto s = load_poly(n, L, moduli, self, PolyRepForm::value);
    s *= scalar;
    store_poly(s, self);
}
void ref_poly_rns_scalar_mul_inplace(size_t n, size_t L, const u64 *moduli, u64 *self,
                                     const u64 *rns_scalar) {
    auto s = load_poly(n, L, moduli, self, PolyRepForm::value);
    s *= mods(rns_scalar, L);
    store_poly(s, self);
}
int ref_poly_ntt(size_t logn, size_t L, const u64 *moduli, u64 *x) {
    return guarded([&] {
        auto p = load_poly((size_t)1 << logn, L, moduli, x, PolyRepForm::coeff);
        ntt_negacyclic_inplace_lazy(p);
        store_poly(p, x);
    });
}
int ref_poly_intt(size_t logn, size_t L, const u64 *moduli, u64 *x) {
    return guarded([&] {
        auto p = load_poly((size_t)1 << logn, L, moduli, x, PolyRepForm::value);
        intt_negacyclic_inplace_lazy(p);
        store_poly(p, x);
    });
}
void ref_poly_reduce_strict(size_t n, size_t L, const u64 *moduli, u64 *x) {
    auto p = load_poly(n, L, moduli, x, PolyRepForm::value);
    reduce_strict(p);
    store_poly(p, x);
}

void ref_poly_involution(size_t logn, size_t L, const u64 *in, u64 *out) {
    std::vector<u64> fake(L, 65537); /* the moduli are irrelevant to the gather */
    store_poly(involution(load_poly((size_t)1 << logn, L, fake.data(), in, PolyRepForm::value)), out);
}
void ref_poly_cycle(size_t logn, size_t L, size_t step, const u64 *in, u64 *out) {
    std::vector<u64> fake(L, 65537);
    store_poly(cycle(load_poly((size_t)1 << logn, L, fake.data(), in, PolyRepForm::value), step), out);
}

int ref_ext_prod_montgomery(size_t logn, size_t L, const u64 *moduli_ext, const u64 *pt,
                            const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        auto p = load_poly(n, L, moduli_ext, pt, PolyRepForm::value);
        auto ct = ext_prod_montgomery(p, load_key(n, L, moduli_ext, key));
        store_poly(ct[0], out);
        store_poly(ct[1], out + (L + 1) * n);
    });
}

int ref_ckks_rescale_by_one_prime(size_t logn, size_t L, const u64 *moduli, const u64 *ct,
                                  u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        CkksCt c;
        c[0] = load_poly(n, L, moduli, ct, PolyRepForm::value);
        c[1] = load_poly(n, L, moduli, ct + L * n, PolyRepForm::value);
        ckks::rescale_inplace(c);
        store_poly(c[0], out);
        store_poly(c[1], out + (L - 1) * n);
    });
}

int ref_bgv_mod_drop_one_prime(size_t logn, size_t L, const u64 *moduli, u64 t, const u64 *ct,
                               u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        bgv::BgvCt c;
        c[0] = load_poly(n, L, moduli, ct, PolyRepForm::value);
        c[1] = load_poly(n, L, moduli, ct + L * n, PolyRepForm::value);
        c.plain_modulus = t;
        bgv::mod_switch_inplace(c);
        store_poly(c[0], out);
        store_poly(c[1], out + (L - 1) * n);
    });
}

void ref_mult_low_level(size_t n, size_t L, const u64 *moduli, const u64 *ct1, const u64 *ct2,
                        u64 *out) {
    CkksCt a, b;
    for (int h = 0; h < 2; h++) {
        a[h] = load_poly(n, L, moduli, ct1 + h * L * n, PolyRepForm::value);
        b[h] = load_poly(n, L, moduli, ct2 + h * L * n, PolyRepForm::value);
    }
    auto q = ckks::mult_low_level(a, b);
    for (int h = 0; h < 3; h++) store_poly(q[h], out + h * L * n);
}

int ref_ckks_relinearize(size_t logn, size_t L, const u64 *moduli_ext, const u64 *quad,
                         const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        ckks::CkksQuadraticCt q;
        for (int h = 0; h < 3; h++) q[h] = load_poly(n, L, moduli_ext, quad + h * L * n, PolyRepForm::value);
        auto r = ckks::relinearize(q, load_key(n, L, moduli_ext, key));
        store_poly(r[0], out);
        store_poly(r[1], out + L * n);
    });
}

/* inner_t is accepted for signature parity with the oracle; the reference
 * hard-wires 1 (bgv.h:32 via bgv/arith.cpp:72). */
int ref_bgv_relinearize(size_t logn, size_t L, const u64 *moduli_ext, u64 /*inner_t*/,
                        const u64 *quad, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        bgv::BgvQuadraticCt q;
        for (int h = 0; h < 3; h++) q[h] = load_poly(n, L, moduli_ext, quad + h * L * n, PolyRepForm::value);
        q.plain_modulus = 65537;
        auto r = bgv::relinearize(q, load_key(n, L, moduli_ext, key));
        store_poly(r[0], out);
        store_poly(r[1], out + L * n);
    });
}

int ref_ckks_mult_relin_rescale(size_t logn, size_t L, const u64 *moduli_ext, const u64 *ct1,
                                const u64 *ct2, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        CkksCt a, b;
        for (int h = 0; h < 2; h++) {
            a[h] = load_poly(n, L, moduli_ext, ct1 + h * L * n, PolyRepForm::value);
            b[h] = load_poly(n, L, moduli_ext, ct2 + h * L * n, PolyRepForm::value);
        }
        auto r = ckks::mult(a, b, load_key(n, L, moduli_ext, key));
        ckks::rescale_inplace(r);
        store_poly(r[0], out);
        store_poly(r[1], out + (L - 1) * n);
    });
}

static int ref_automorphism(size_t logn, size_t L, const u64 *moduli_ext, bool conj, size_t step, const u64 *ct,
                            const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        CkksCt a;
        for (int h = 0; h < 2; h++) a[h] = load_poly(n, L, moduli_ext, ct + h * L * n, PolyRepForm::value);
        auto ksk = load_key(n, L, moduli_ext, key);
        CkksCt r = conj ? ckks::conjugate(a, ksk) : ckks::rotate(a, ksk, step);
        store_poly(r[0], out);
        store_poly(r[1], out + L * n);
    });
}
int ref_ckks_rotate(size_t logn, size_t L, const u64 *moduli_ext, size_t step, const u64 *ct, const u64 *key, u64 *out) {
    return ref_automorphism(logn, L, moduli_ext, false, step, ct, key, out);
}
int ref_ckks_conjugate(size_t logn, size_t L, const u64 *moduli_ext, const u64 *ct, const u64 *key, u64 *out) {
    return ref_automorphism(logn, L, moduli_ext, true, 0, ct, key, out);
}

// encrypt_core (rlwe.cpp:57-72) draws its samples from the library's private RNG, so the same four lines are
// replayed here on caller-supplied samples with the reference's own operators and transforms
int ref_rlwe_encrypt_core(size_t logn, size_t L, const u64 *moduli, const int64_t *noise, const u64 *c1, const u64 *pt,
                          const u64 *sk, u64 *ct) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        RnsPolynomial ex(n, L, mods(moduli, L));
        for (size_t k = 0; k < L; k++)
            for (size_t i = 0; i < n; i++) {   // sampling.cpp:77-83
                u64 coeff = moduli[k] + (u64)noise[i];
                coeff -= (coeff >= moduli[k]) ? moduli[k] : 0;
                ex[k][i] = coeff;
            }
        ntt_negacyclic_inplace_lazy(ex);                                   // sampling.cpp:86
        auto c1p = load_poly(n, L, moduli, c1, PolyRepForm::value);
        auto skp = load_poly(n, L, moduli, sk, PolyRepForm::value);
        auto c0 = ex - c1p * skp;                                          // rlwe.cpp:52
        auto pt_ntt = load_poly(n, L, moduli, pt, PolyRepForm::coeff);
        ntt_negacyclic_inplace_lazy(pt_ntt);                               // rlwe.cpp:66-67
        c0 += pt_ntt;                                                      // rlwe.cpp:70
        store_poly(c0, ct);
        store_poly(c1p, ct + L * n);
    });
}
int ref_rlwe_decrypt_core(size_t logn, size_t L, const u64 *moduli, const u64 *ct, const u64 *sk, u64 *pt) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        RlweCt c{load_poly(n, L, moduli, ct, PolyRepForm::value), load_poly(n, L, moduli, ct + L * n, PolyRepForm::value)};
        RlweSk s(n, L, mods(moduli, L));
        for (size_t k = 0; k < L; k++) std::memcpy(s[k].data(), sk + k * n, n * sizeof(u64));
        s.rep_form = PolyRepForm::value;
        store_poly(decrypt_core(c, s), pt);
    });
}
void ref_rns_base_from_single(size_t n, u64 old_modulus, size_t L, const u64 *new_moduli, const u64 *in, u64 *out) {
    u64 m = old_modulus;
    auto p = load_poly(n, 1, &m, in, PolyRepForm::coeff);
    store_poly(rns_base_transform(p, mods(new_moduli, L)), out);
}
void ref_rns_base_to_single(size_t n, size_t L, const u64 *old_moduli, u64 new_modulus, const u64 *in, u64 *out) {
    auto p = load_poly(n, L, old_moduli, in, PolyRepForm::coeff);
    store_poly(rns_base_transform(p, std::vector<u64>{new_modulus}), out);
}
int ref_rns_base_to_single_small(size_t n, size_t L, const u64 *old_moduli, u64 new_modulus, const u64 *in, u64 *out) {
    auto p = load_poly(n, L, old_moduli, in, PolyRepForm::coeff);
    store_poly(rns_base_transform(p, std::vector<u64>{new_modulus}), out);   // takes the CRT branch by itself when needed
    return 1;
}

int ref_bgv_mult_relin_modswitch(size_t logn, size_t L, const u64 *moduli_ext, u64 t,
                                 const u64 *ct1, const u64 *ct2, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    return guarded([&] {
        bgv::BgvCt a, b;
        for (int h = 0; h < 2; h++) {
            a[h] = load_poly(n, L, moduli_ext, ct1 + h * L * n, PolyRepForm::value);
            b[h] = load_poly(n, L, moduli_ext, ct2 + h * L * n, PolyRepForm::value);
        }
        a.plain_modulus = b.plain_modulus = t;
        auto r = bgv::relinearize(bgv::mult_low_level(a, b), load_key(n, L, moduli_ext, key));
        bgv::mod_switch_inplace(r);
        store_poly(r[0], out);
        store_poly(r[1], out + (L - 1) * n);
    });
}

/* ---- timing loops (cpu_baseline "reference") ------------------------- */

/* seconds per forward (inverse=0) or inverse limb transform, tables warm */
double ref_time_ntt(size_t logn, u64 q, int inverse, size_t iters, u64 *x) {
    if (inverse) intt_negacyclic_inplace_lazy(logn, q, x); else ntt_negacyclic_inplace_lazy(logn, q, x);
    const double t0 = now_s();
    for (size_t it = 0; it < iters; it++) {
        if (inverse) intt_negacyclic_inplace_lazy(logn, q, x); else ntt_negacyclic_inplace_lazy(logn, q, x);
    }
    return (now_s() - t0) / (double)iters;
}

/* seconds per ckks::mult + rescale_inplace on reference containers */
double ref_time_ckks_mult(size_t logn, size_t L, const u64 *moduli_ext, const u64 *ct1,
                          const u64 *ct2, const u64 *key, size_t iters) {
    const size_t n = (size_t)1 << logn;
    CkksCt a, b;
    for (int h = 0; h < 2; h++) {
        a[h] = load_poly(n, L, moduli_ext, ct1 + h * L * n, PolyRepForm::value);
        b[h] = load_poly(n, L, moduli_ext, ct2 + h * L * n, PolyRepForm::value);
    }
    auto ksk = load_key(n, L, moduli_ext, key);
    {
        auto r = ckks::mult(a, b, ksk);
        ckks::rescale_inplace(r);
    }
    const double t0 = now_s();
    for (size_t it = 0; it < iters; it++) {
        auto r = ckks::mult(a, b, ksk);
        ckks::rescale_inplace(r);
    }
    return (now_s() - t0) / (double)iters;
}

double ref_time_bgv_mult(size_t logn, size_t L, const u64 *moduli_ext, u64 t, const u64 *ct1,
                         const u64 *ct2, const u64 *key, size_t iters) {
    const size_t n = (size_t)1 << logn;
    bgv::BgvCt a, b;
    for (int h = 0; h < 2; h++) {
        a[h] = load_poly(n, L, moduli_ext, ct1 + h * L * n, PolyRepForm::value);
        b[h] = load_poly(n, L, moduli_ext, ct2 + h * L * n, PolyRepForm::value);
    }
    a.plain_modulus = b.plain_modulus = t;
    auto ksk = load_key(n, L, moduli_ext, key);
    const double t0 = now_s();
    for (size_t it = 0; it < iters; it++) {
        auto r = bgv::relinearize(bgv::mult_low_level(a, b), ksk);
        bgv::mod_switch_inplace(r);
    }
    return (now_s() - t0) / (double)iters;
}

} // extern "C"
