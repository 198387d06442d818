// hehub.cpp -- implementation of the hehub-compatible host layer over the C ABI.  No arithmetic on
// ring elements happens in this file: it validates arguments the way the reference does, stages host
// limbs into contiguous device batches, calls the engine and copies the results back.
//
// Two ways to build it:
//   default                      against hehub.hpp, our own mirror of the reference's types (hehub_amd/host);
//   -DHEHUB_AMD_BIND_REFERENCE   against the reference's OWN headers (-I<hehub>/src): the file then defines
//                                only the functions hehub defines out of line on the hot path (the ones listed in
//                                SURVEY.md section 8a), with hehub's exact signatures, so linking it ahead of
//                                hehub's ntt.cpp / mod_arith.cpp / rns.cpp / rgsw.cpp / rescaling.cpp /
//                                mod_switch.cpp / arith.cpp / permutation.cpp moves that path to the GPU while
//                                everything else (sampling, encoding, key generation, circuits, tests) stays
//                                hehub's.  The `ref_tests` make target does exactly that with hehub's
//                                own test suite.
#ifdef HEHUB_AMD_BIND_REFERENCE
#include "fhe/bgv/bgv.h"
#include "fhe/ckks/ckks.h"
#include "fhe/common/mod_arith.h"
#include "fhe/common/ntt.h"
#include "fhe/common/permutation.h"
#include "fhe/common/rns.h"
#include "fhe/primitives/keys.h"
#include "fhe/primitives/rgsw.h"
#include "fhe/primitives/rlwe.h"
struct hp_ctx;
namespace hehub { namespace amd { hp_ctx *engine(); } }
#else
#include "hehub.hpp"
#endif

#include "../../include/hehub_amd.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>

namespace hehub {

namespace amd {

hp_ctx *engine() {
    static hp_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        if (const char *e = std::getenv("HEHUB_AMD_DEVICE")) dev = std::atoi(e);
        if (hp_ctx_create(dev, &ctx) != HP_OK) ctx = nullptr;
    });
    if (!ctx) throw std::runtime_error("hehub_amd: no MI355X engine available (hp_ctx_create failed); there is no CPU fallback");
    return ctx;
}

} // namespace amd

namespace {

// number of engine calls made through this layer; printed at exit when HEHUB_AMD_VERBOSE is set, so a run of
// somebody else's test-suite over this layer can show that the work really went to the device
struct CallCounter {
    unsigned long long n = 0;
    ~CallCounter() {
        if (std::getenv("HEHUB_AMD_VERBOSE")) std::fprintf(stderr, "hehub_amd: %llu engine calls (%s)\n", n, hp_version());
    }
} g_calls;

void check(int rc) {
    g_calls.n++;
    if (rc == HP_OK) return;
    std::string msg = hp_last_error(amd::engine());   // the calling thread's own last failure (hp_ctx.cpp)
    if (rc == HP_EINVAL) throw std::invalid_argument(msg);
    if (rc == HP_ELOGIC) throw std::logic_error(msg);
    throw std::runtime_error("hehub_amd: " + msg);
}

// RAII device buffer of `words` u64
struct DevBuf {
    u64 *p = nullptr;
    explicit DevBuf(size_t words) {
        void *d = nullptr;
        check(hp_dev_alloc(amd::engine(), words * sizeof(u64), &d));
        p = (u64 *)d;
    }
    ~DevBuf() {
        if (p) hp_dev_free(amd::engine(), p);
    }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

void put_poly(u64 *dst, const RnsIntVec &v, size_t limbs) {
    const size_t n = v.dimension();
    for (size_t k = 0; k < limbs; k++) check(hp_memcpy_h2d(amd::engine(), dst + k * n, v[(int)k].data(), n * sizeof(u64)));
}

void get_poly(RnsIntVec &v, const u64 *src, size_t limbs) {
    const size_t n = v.dimension();
    for (size_t k = 0; k < limbs; k++) check(hp_memcpy_d2h(amd::engine(), v[(int)k].data(), src + k * n, n * sizeof(u64)));
}

// rns.cpp:59-72 shared precondition of += and -=
size_t check_addsub(const RnsIntVec &self, const RnsIntVec &b) {
    if (self.dimension() != b.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    if (b.component_count() < self.component_count())
        throw std::invalid_argument("Operand b contains less components than self.");
    auto components = self.component_count();
    auto moduli(self.modulus_vec()), b_moduli(b.modulus_vec());
    b_moduli.resize(components);
    if (moduli != b_moduli) throw std::invalid_argument("Operands' moduli mismatch.");
    return components;
}

enum class Bin { add, sub, mul };

void run_binary(Bin op, const RnsIntVec &a, const RnsIntVec &b, RnsIntVec &out, size_t L) {
    const size_t n = a.dimension();
    if (L == 0 || n == 0) return;
    DevBuf da(L * n), db(L * n);
    put_poly(da.p, a, L);
    put_poly(db.p, b, L);
    auto *ctx = amd::engine();
    const u64 *m = a.modulus_vec().data();
    if (op == Bin::add) check(hp_dev_poly_add(ctx, n, L, m, 1, da.p, db.p, da.p));
    if (op == Bin::sub) check(hp_dev_poly_sub(ctx, n, L, m, 1, da.p, db.p, da.p));
    if (op == Bin::mul) check(hp_dev_poly_mul(ctx, n, L, m, 1, da.p, db.p, da.p));
    get_poly(out, da.p, L);
}

void scalar_mul(RnsIntVec &self, const std::vector<u64> &scalars) {
    const size_t n = self.dimension(), L = self.component_count();
    if (L == 0) return;
    DevBuf d(L * n);
    put_poly(d.p, self, L);
    check(hp_dev_poly_scalar_mul(amd::engine(), n, L, self.modulus_vec().data(), 1, scalars.data(), d.p, d.p));
    get_poly(self, d.p, L);
}

void check_ct_wellformed(const RlweCt &ct) {   // rescaling.cpp:15-29, mod_switch.cpp:14-28
    if (ct[0].modulus_vec() != ct[1].modulus_vec())
        throw std::invalid_argument("Ill-formed ciphertext: modulus sets mismatch.");
    if (ct[0].dimension() != ct[1].dimension())
        throw std::invalid_argument("Ill-formed ciphertext: polynomial lengths mismatch.");
    if (ct[0].component_count() != ct[1].component_count())
        throw std::invalid_argument("Ill-formed ciphertext: component numbers mismatch.");
    if (ct[0].component_count() == 1) throw std::invalid_argument("Unable to drop the only one prime.");
}

// HEHUB_AMD_EXTENSIONS=1 turns on what hehub itself throws for (include/hehub_amd.h "extensions"): a key generated
// for more ciphertext moduli than the operand has, and rescale_inplace by several primes
bool extensions_on() {
    static const bool on = std::getenv("HEHUB_AMD_EXTENSIONS") && std::atoi(std::getenv("HEHUB_AMD_EXTENSIONS")) != 0;
    return on;
}

// rgsw.cpp:58-89; returns the number of ciphertext moduli the key was generated for (== pt's unless extensions are on)
size_t check_ext_prod(const RlwePt &pt, const RgswCt &rgsw, std::vector<u64> &extended_moduli) {
    if (rgsw.empty()) throw std::invalid_argument("Empty RGSW ciphertext.");
    extended_moduli = rgsw[0][0].modulus_vec();
    const auto original = pt.component_count();
    const auto extended = original + 1;
    if (extended_moduli.size() < extended) throw std::invalid_argument("Invalid component number in RGSW ciphertext.");
    const std::vector<u64> key_moduli(extended_moduli);
    extended_moduli.resize(extended);
    *extended_moduli.rbegin() = *rgsw[0][0].modulus_vec().crbegin();
    for (size_t i = 0; i < original; i++)
        if (extended_moduli[i] != pt.modulus_at((int)i)) throw std::invalid_argument("Moduli mismatch.");
    const bool higher = extensions_on() && rgsw.size() > original;
    for (auto &sample : rgsw)
        for (auto &poly : sample) {
            if (poly.dimension() != pt.dimension()) throw std::invalid_argument("Polynomial lengths mismatch.");
            if (higher ? (poly.component_count() != rgsw.size() + 1 || poly.modulus_vec() != key_moduli)
                       : (poly.component_count() != extended || poly.modulus_vec() != extended_moduli))
                throw std::invalid_argument("Inconsistent RGSW ciphertext.");
        }
    if (!higher && rgsw.size() != original) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    return rgsw.size();
}

void put_key(u64 *dst, const RgswCt &rgsw, size_t L, size_t n) {
    for (size_t j = 0; j < L; j++)
        for (size_t h = 0; h < 2; h++) put_poly(dst + ((j * 2 + h) * (L + 1)) * n, rgsw[j][h], L + 1);
}

// Device copy of a key-switching key for one call.  A key is 2L(L+1) limbs (55 MiB at N=32768, L=10) and is the same
// object call after call, so staging it every time dominates the host-pointer path.  With HEHUB_AMD_KEY_CACHE=<entries>
// the layer keeps up to that many keys resident, recognised by the address of their first limb, their shape and four
// sampled words of every limb (a key that is modified in place between calls without touching any sampled word would go
// unnoticed: that is why the cache is opt-in).  Default: no cache, the key is staged per call.
class DevKey {
public:
    DevKey(const RgswCt &rgsw, size_t L, size_t n) {
        static const size_t cap = std::getenv("HEHUB_AMD_KEY_CACHE") ? (size_t)std::atoi(std::getenv("HEHUB_AMD_KEY_CACHE")) : 0;
        const size_t words = L * 2 * (L + 1) * n;
        if (cap == 0) {
            own_.reset(new DevBuf(words));
            put_key(own_->p, rgsw, L, n);
            p_ = own_->p;
            return;
        }
        std::vector<u64> sig{(u64)(uintptr_t)rgsw[0][0][0].data(), (u64)L, (u64)n};
        for (size_t j = 0; j < L; j++)
            for (size_t h = 0; h < 2; h++)
                for (size_t k = 0; k <= L; k++) {
                    const u64 *w = rgsw[j][h][(int)k].data();
                    sig.insert(sig.end(), {w[0], w[n / 3], w[(2 * n) / 3], w[n - 1]});
                }
        // Heap-allocated and never destroyed on purpose: a static object's destructor would run hipFree during static
        // destruction at process exit, when the HIP runtime may already be gone (crash or hang at exit).  The cached
        // device blocks go back with the process.
        typedef std::vector<std::pair<std::vector<u64>, std::shared_ptr<DevBuf>>> Cache;   // most recently used last
        static std::mutex &mu = *new std::mutex;
        static Cache &cache = *new Cache;
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < cache.size(); i++)
            if (cache[i].first == sig) {
                auto hit = cache[i];
                cache.erase(cache.begin() + i);
                cache.push_back(hit);
                own_ = hit.second;
                p_ = own_->p;
                return;
            }
        own_.reset(new DevBuf(words));
        put_key(own_->p, rgsw, L, n);
        p_ = own_->p;
        if (cache.size() >= cap) cache.erase(cache.begin());
        cache.emplace_back(std::move(sig), own_);
    }
    const u64 *p() const { return p_; }

private:
    std::shared_ptr<DevBuf> own_;
    u64 *p_ = nullptr;
};

RlweCt make_ct(size_t n, size_t L, const std::vector<u64> &moduli, const u64 *src) {
    RlweCt ct{RnsPolynomial(n, L, moduli), RnsPolynomial(n, L, moduli)};
    for (int h = 0; h < 2; h++) {
        get_poly(ct[h], src + (size_t)h * L * n, L);
        ct[h].rep_form = PolyRepForm::value;
    }
    return ct;
}

// shared body of ckks::relinearize / bgv::relinearize
RlweCt relinearize_common(const std::array<RnsPolynomial, 3> &quad, const RlweKsk &key, bool bgv) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(quad[2], key, mext);
    const size_t n = quad[2].dimension(), L = quad[2].component_count();
    DevBuf dq(3 * L * n), dout(2 * L * n);
    DevKey dk(key, L0, n);
    for (int h = 0; h < 3; h++) put_poly(dq.p + (size_t)h * L * n, quad[h], L);
    const size_t logn = quad[2].log_dimension();
    if (bgv && L0 != L) throw std::invalid_argument("Inconsistent RGSW ciphertext.");   // no higher-level keys for the BGV quirk path
    if (bgv) check(hp_dev_bgv_relinearize(amd::engine(), logn, L, mext.data(), 1 /* bgv.h:32 */, 1, dq.p, dk.p(), dout.p));
    else check(hp_dev_ckks_relinearize_at(amd::engine(), logn, L, L0, mext.data(), 1, dq.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    return make_ct(n, L, q, dout.p);
}

template <class Quad, class Ct> Quad mult_low_level_common(const Ct &ct1, const Ct &ct2) {
    for (int h = 0; h < 2; h++) {
        if (ct1[h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
        if (ct2[h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    }
    if (ct1[0].dimension() != ct2[0].dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t n = ct1[0].dimension();
    const size_t L = std::min(ct1[0].component_count(), ct2[0].component_count());
    std::vector<u64> m1(ct1[0].modulus_vec()), m2(ct2[0].modulus_vec());
    m1.resize(L); m2.resize(L);
    if (m1 != m2) throw std::invalid_argument("Operands' moduli mismatch.");
    DevBuf d1(2 * L * n), d2(2 * L * n), dq(3 * L * n);
    for (int h = 0; h < 2; h++) {
        put_poly(d1.p + (size_t)h * L * n, ct1[h], L);
        put_poly(d2.p + (size_t)h * L * n, ct2[h], L);
    }
    check(hp_dev_mult_low_level(amd::engine(), ct1[0].log_dimension(), L, m1.data(), 1, d1.p, d2.p, dq.p));
    Quad quad;
    for (int h = 0; h < 3; h++) {
        quad[h] = RnsPolynomial(n, L, m1);
        get_poly(quad[h], dq.p + (size_t)h * L * n, L);
        quad[h].rep_form = PolyRepForm::value;
    }
    return quad;
}

void drop_last_prime(RlweCt &ct, bool bgv, u64 t) {
    check_ct_wellformed(ct);
    const size_t n = ct[0].dimension(), L = ct[0].component_count(), logn = ct[0].log_dimension();
    DevBuf din(2 * L * n), dout(2 * (L - 1) * n);
    for (int h = 0; h < 2; h++) put_poly(din.p + (size_t)h * L * n, ct[h], L);
    const u64 *m = ct[0].modulus_vec().data();
    if (bgv) check(hp_dev_bgv_mod_switch(amd::engine(), logn, L, m, t, 1, din.p, dout.p));
    else check(hp_dev_ckks_rescale(amd::engine(), logn, L, m, 1, din.p, dout.p));
    for (int h = 0; h < 2; h++) {
        ct[h].remove_components();
        get_poly(ct[h], dout.p + (size_t)h * (L - 1) * n, L - 1);
    }
}

} // namespace

// =====================================================================================================
// rns.h / rns.cpp
// =====================================================================================================
#ifndef HEHUB_AMD_BIND_REFERENCE
RnsIntVec::RnsIntVec(size_t dimension, size_t components, const std::vector<u64> &moduli) {
    size_t lg = 0;
    while (((size_t)1 << lg) < dimension) lg++;
    if (dimension == 0 || dimension != (size_t)1 << lg) throw std::invalid_argument("dimension should be a 2-power.");   // rns.cpp:17-22
    if (moduli.size() < components) throw std::invalid_argument("No matching number of moduli provided to create RnsIntVec.");
    logn_ = lg;
    q_.assign(moduli.begin(), moduli.begin() + components);
    limbs_.assign(components, ComponentData(dimension));
}

RnsIntVec::RnsIntVec(const RnsIntVec::Params &p) : RnsIntVec(p.dimension, p.component_count, p.moduli) {}

void RnsIntVec::add_components(const std::vector<u64> &new_moduli, size_t adding) {
    if (new_moduli.size() < adding) throw std::invalid_argument("No matching number of moduli provided to add components.");
    q_.insert(q_.end(), new_moduli.begin(), new_moduli.end());   // rns.cpp:41: every supplied modulus is appended, `adding` limbs are
    limbs_.insert(limbs_.end(), adding, ComponentData(dimension()));
}

void RnsIntVec::remove_components(size_t removing) {
    if (component_count() < removing) throw std::invalid_argument("Trying to remove components more than existing.");
    q_.resize(q_.size() - removing);
    limbs_.resize(limbs_.size() - removing);
}
#endif

const RnsIntVec &operator+=(RnsIntVec &self, const RnsIntVec &b) {
    run_binary(Bin::add, self, b, self, check_addsub(self, b));
    return self;
}

const RnsIntVec &operator-=(RnsIntVec &self, const RnsIntVec &b) {
    run_binary(Bin::sub, self, b, self, check_addsub(self, b));
    return self;
}

RnsIntVec operator*(const RnsIntVec &a, const RnsIntVec &b) {
    if (a.dimension() != b.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    auto components = std::min(a.component_count(), b.component_count());
    auto moduli(a.modulus_vec()), b_moduli(b.modulus_vec());
    moduli.resize(components);
    b_moduli.resize(components);
    if (moduli != b_moduli) throw std::invalid_argument("Operands' moduli mismatch.");
    RnsIntVec result(RnsIntVec::Params{a.dimension(), components, moduli});
    run_binary(Bin::mul, a, b, result, components);
    return result;
}

const RnsIntVec &operator*=(RnsIntVec &self, const u64 small_scalar) {
    scalar_mul(self, std::vector<u64>(self.component_count(), small_scalar));
    return self;
}

const RnsIntVec &operator*=(RnsIntVec &self, const std::vector<u64> &rns_scalar) {
    if (rns_scalar.size() != self.component_count()) throw std::invalid_argument("Numbers of RNS component mismatch.");
    scalar_mul(self, rns_scalar);
    return self;
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // inline in the reference's rns.h:207-293
const RnsPolynomial &operator+=(RnsPolynomial &self, const RnsPolynomial &b) {
    if (self.rep_form != b.rep_form) throw std::invalid_argument("Operands are in different representation form.");
    (RnsIntVec &)self += (const RnsIntVec &)b;
    return self;
}

const RnsPolynomial &operator-=(RnsPolynomial &self, const RnsPolynomial &b) {
    if (self.rep_form != b.rep_form) throw std::invalid_argument("Operands are in different representation form.");
    (RnsIntVec &)self -= (const RnsIntVec &)b;
    return self;
}

RnsPolynomial operator*(const RnsPolynomial &a, const RnsPolynomial &b) {
    if (a.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
    if (b.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    RnsPolynomial result = (const RnsIntVec &)a * (const RnsIntVec &)b;
    result.rep_form = PolyRepForm::value;
    return result;
}

const RnsPolynomial &operator*=(RnsPolynomial &self, const u64 s) {
    (RnsIntVec &)self *= s;
    return self;
}

const RnsPolynomial &operator*=(RnsPolynomial &self, const std::vector<u64> &s) {
    (RnsIntVec &)self *= s;
    return self;
}
#endif

// =====================================================================================================
// mod_arith.h
// =====================================================================================================
void batched_barrett_lazy(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett_lazy(amd::engine(), q, n, v)); }
#ifndef HEHUB_AMD_BIND_REFERENCE   // inline in the reference's mod_arith.h:18-25,58-63
void batched_barrett(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett(amd::engine(), q, n, v)); }
void batched_reduce_strict(const u64 q, const size_t n, u64 v[]) { check(hp_batched_reduce_strict(amd::engine(), q, n, v)); }
#endif
void batched_mul_mod_hybrid_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_hybrid_lazy(amd::engine(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_hybrid(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_hybrid_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_mul_mod_barrett_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_barrett_lazy(amd::engine(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_barrett(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_barrett_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_montgomery_128_lazy(const u64 q, const size_t len, const u128 in[], u64 out[]) {
    check(hp_batched_montgomery_128_lazy(amd::engine(), q, len, reinterpret_cast<const u64 *>(in), out));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // mod_arith.h:65-72 (inline) and mod_arith.cpp:136-149 stay the reference's
void reduce_strict(RnsPolynomial &p) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L == 0) return;
    DevBuf d(L * n);
    put_poly(d.p, p, L);
    check(hp_dev_poly_reduce_strict(amd::engine(), n, L, p.modulus_vec().data(), 1, d.p));
    get_poly(p, d.p, L);
}

// host-side scalar, as in the reference (mod_arith.cpp:136-149): Bezout coefficient lifted to [0, prime)
u64 inverse_mod_prime(const u64 elem, const u64 prime) {
    __int128 r0 = prime, r1 = elem, y0 = 0, y1 = 1;
    while (r1 != 0) {
        __int128 quo = r0 / r1, r2 = r0 - quo * r1, y2 = y0 - quo * y1;
        r0 = r1; r1 = r2; y0 = y1; y1 = y2;
    }
    if (y0 < 0) y0 += prime;
    return (u64)y0;
}
#endif

// =====================================================================================================
// ntt.h
// =====================================================================================================
void ntt_negacyclic_inplace_lazy(const size_t logn, const u64 q, u64 c[]) {
    check(hp_ntt_negacyclic_inplace_lazy(amd::engine(), logn, q, c));
}
void intt_negacyclic_inplace_lazy(const size_t logn, const u64 q, u64 v[]) {
    check(hp_intt_negacyclic_inplace_lazy(amd::engine(), logn, q, v));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // ntt.h:41-92 (inline per-limb loops) stay the reference's
static void poly_transform(RnsPolynomial &p, bool inverse, bool strict) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L) {
        DevBuf d(L * n);
        put_poly(d.p, p, L);
        if (inverse) check(hp_dev_intt(amd::engine(), p.log_dimension(), L, p.modulus_vec().data(), 1, d.p, strict ? 1 : 0));
        else check(hp_dev_ntt(amd::engine(), p.log_dimension(), L, p.modulus_vec().data(), 1, d.p));
        get_poly(p, d.p, L);
    }
    p.rep_form = inverse ? PolyRepForm::coeff : PolyRepForm::value;
}
void ntt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, false, false); }
void intt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, true, false); }
void intt_negacyclic_inplace(RnsPolynomial &p) { poly_transform(p, true, true); }
#endif

void cache_ntt_factors_strict(const u64 logn, const std::vector<u64> &moduli) {
    check(hp_cache_ntt_factors_strict(amd::engine(), logn, moduli.data(), moduli.size()));
}

// =====================================================================================================
// permutation.h
// =====================================================================================================
static RnsPolynomial gather(const RnsPolynomial &p, bool is_cycle, size_t step) {
    if (p.rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    const size_t n = p.dimension(), L = p.component_count();
    RnsPolynomial out(n, L, p.modulus_vec());
    out.rep_form = PolyRepForm::value;
    if (L == 0) return out;
    DevBuf din(L * n), dout(L * n);
    put_poly(din.p, p, L);
    if (is_cycle) check(hp_dev_poly_cycle(amd::engine(), p.log_dimension(), L, 1, step, din.p, dout.p));
    else check(hp_dev_poly_involution(amd::engine(), p.log_dimension(), L, 1, din.p, dout.p));
    get_poly(out, dout.p, L);
    return out;
}
RnsPolynomial cycle(const RnsPolynomial &p, const size_t step) { return gather(p, true, step); }
RnsPolynomial involution(const RnsPolynomial &p) { return gather(p, false, 0); }

// =====================================================================================================
// rlwe.h / rgsw.h
// =====================================================================================================
#ifndef HEHUB_AMD_BIND_REFERENCE   // rlwe.cpp:83-101: thin compositions of the operators above
RlweCt add(const RlweCt &a, const RlweCt &b) { return RlweCt{a[0] + b[0], a[1] + b[1]}; }
RlweCt sub(const RlweCt &a, const RlweCt &b) { return RlweCt{a[0] - b[0], a[1] - b[1]}; }
RlweCt add_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] + pt, ct[1]}; }
RlweCt sub_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] - pt, ct[1]}; }
RlweCt mult_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] * pt, ct[1] * pt}; }
#endif

RlweCt ext_prod_montgomery(const RlwePt &pt, const RgswCt &rgsw) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(pt, rgsw, mext);
    const size_t n = pt.dimension(), L = pt.component_count();
    DevBuf dp(L * n), dout(2 * (L + 1) * n);
    DevKey dk(rgsw, L0, n);
    put_poly(dp.p, pt, L);
    check(hp_dev_ext_prod_montgomery_at(amd::engine(), pt.log_dimension(), L, L0, mext.data(), 1, dp.p, dk.p(), dout.p));
    return make_ct(n, L + 1, mext, dout.p);
}

// rlwe.h decrypt_core (rlwe.cpp:74-81): `c0 + c1 * sk`, INTT, reduce_strict as ONE device call instead of 3L host
// round trips; the argument checks are the ones the reference's operator* / operator+ perform, in their order.
RlwePt decrypt_core(const RlweCt &ct, const RlweSk &sk) {
    const RnsPolynomial &c0 = ct[0], &c1 = ct[1];
    if (c1.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
    if (sk.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    if (c1.dimension() != sk.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t Lp = std::min(c1.component_count(), sk.component_count());
    std::vector<u64> m1(c1.modulus_vec()), ms(sk.modulus_vec());
    m1.resize(Lp); ms.resize(Lp);
    if (m1 != ms) throw std::invalid_argument("Operands' moduli mismatch.");
    if (c0.rep_form != PolyRepForm::value) throw std::invalid_argument("Operands are in different representation form.");
    if (c0.dimension() != c1.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t L = c0.component_count(), n = c0.dimension();
    if (Lp < L) throw std::invalid_argument("Operand b contains less components than self.");
    m1.resize(L);
    if (c0.modulus_vec() != m1) throw std::invalid_argument("Operands' moduli mismatch.");
    RnsPolynomial pt(n, L, m1);
    pt.rep_form = PolyRepForm::coeff;
    if (L == 0) return pt;
    DevBuf dct(2 * L * n), dsk(L * n), dpt(L * n);
    put_poly(dct.p, c0, L);
    put_poly(dct.p + L * n, c1, L);
    put_poly(dsk.p, sk, L);
    check(hp_dev_rlwe_decrypt_core(amd::engine(), c0.log_dimension(), L, m1.data(), 1, dct.p, dsk.p, dpt.p));
    get_poly(pt, dpt.p, L);
    return pt;
}

#ifndef HEHUB_AMD_BIND_REFERENCE
// rns_transform.cpp:106-127 on the device: one -> many (:11-37) and many -> one (:39-104, both branches)
RnsPolynomial rns_base_transform(RnsPolynomial in, const std::vector<u64> &new_moduli) {
    if (in.rep_form == PolyRepForm::value)
        throw std::logic_error("Trying to perform RNS base transformation on NTT values.");
    const size_t n = in.dimension(), L = in.component_count();
    if (L == 1) {
        RnsPolynomial out(n, new_moduli.size(), new_moduli);
        if (new_moduli.empty()) return out;
        DevBuf din(n), dout(new_moduli.size() * n);
        put_poly(din.p, in, 1);
        check(hp_dev_rns_base_from_single(amd::engine(), n, in.modulus_at(0), new_moduli.size(), new_moduli.data(), 1, din.p, dout.p));
        get_poly(out, dout.p, new_moduli.size());
        return out;
    }
    if (new_moduli.size() == 1) {   // both branches of rns_transform.cpp:39-104 on the device
        RnsPolynomial out(n, 1, new_moduli);
        DevBuf din(L * n), dout(n);
        put_poly(din.p, in, L);
        check(hp_dev_rns_base_to_single(amd::engine(), n, L, in.modulus_vec().data(), new_moduli[0], 1, din.p, dout.p));
        get_poly(out, dout.p, 1);
        return out;
    }
    throw "under development";   // rns_transform.cpp:123
}
#endif

// =====================================================================================================
// ckks.h
// =====================================================================================================
namespace ckks {

#ifndef HEHUB_AMD_BIND_REFERENCE   // ckks/arith.cpp:7-53 stay the reference's
static void check_scaling_factor(double a, double b) {   // ckks/arith.cpp:7-13
    if (std::abs(a - b) > std::pow(2.0, -50)) throw std::invalid_argument("The scaling factors mismatch");
}

CkksCt add(const CkksCt &a, const CkksCt &b) {
    check_scaling_factor(a.scaling_factor, b.scaling_factor);
    CkksCt r = ::hehub::add((const RlweCt &)a, (const RlweCt &)b);
    r.scaling_factor = a.scaling_factor;
    return r;
}

CkksCt sub(const CkksCt &a, const CkksCt &b) {
    check_scaling_factor(a.scaling_factor, b.scaling_factor);
    CkksCt r = ::hehub::sub((const RlweCt &)a, (const RlweCt &)b);
    r.scaling_factor = a.scaling_factor;
    return r;
}

CkksCt add_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:22-29
    check_scaling_factor(ct.scaling_factor, pt.scaling_factor);
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = add_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt sub_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:38-45
    check_scaling_factor(ct.scaling_factor, pt.scaling_factor);
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = sub_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt mult_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:47-53
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = mult_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor * pt.scaling_factor;
    return r;
}
#endif

CkksQuadraticCt mult_low_level(const CkksCt &a, const CkksCt &b) {
    auto q = mult_low_level_common<CkksQuadraticCt>(a, b);
    q.scaling_factor = a.scaling_factor * b.scaling_factor;
    return q;
}

CkksCt relinearize(const CkksQuadraticCt &ct, const RlweKsk &key) {
    CkksCt r = relinearize_common(ct, key, false);
    r.scaling_factor = ct.scaling_factor;   // ckks/arith.cpp:68
    return r;
}

void rescale_inplace(CkksCt &ct, size_t dropping_primes) {   // rescaling.cpp:80-90
    if (dropping_primes == 1) {
        check_ct_wellformed(ct);
        const u64 q_last = *ct[0].modulus_vec().crbegin();
        drop_last_prime(ct, false, 0);
        ct.scaling_factor /= q_last;
    } else if (dropping_primes >= 2) {
        if (!extensions_on()) throw "under development";
        for (size_t d = 0; d < dropping_primes; d++) rescale_inplace(ct, 1);   // successive exact one-prime drops
    } else {
        throw std::invalid_argument("The number of primes to be dropped is not positive.");
    }
}

// ckks/arith.cpp:75-93: automorphism, key switch, drop of the special prime and the add of moved[0] run as one
// device call; the argument checks below are the ones the reference's composition performs, in its order.
static CkksCt key_switched(const CkksCt &ct, const RlweKsk &key, bool conj, size_t step) {
    for (int h = 0; h < 2; h++)
        if (ct[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(ct[1], key, mext);
    const size_t n = ct[1].dimension(), L = ct[1].component_count(), logn = ct[1].log_dimension();
    if (ct[0].dimension() != n || ct[0].component_count() != L) throw std::invalid_argument("Ill-formed ciphertext.");
    DevBuf dct(2 * L * n), dout(2 * L * n);
    DevKey dk(key, L0, n);
    for (int h = 0; h < 2; h++) put_poly(dct.p + (size_t)h * L * n, ct[h], L);
    if (conj) check(hp_dev_ckks_conjugate_at(amd::engine(), logn, L, L0, mext.data(), 1, dct.p, dk.p(), dout.p));
    else check(hp_dev_ckks_rotate_at(amd::engine(), logn, L, L0, mext.data(), 1, step, dct.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    CkksCt r = make_ct(n, L, q, dout.p);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt conjugate(const CkksCt &ct, const RlweKsk &conj_key) { return key_switched(ct, conj_key, true, 0); }

CkksCt rotate(const CkksCt &ct, const RlweKsk &rot_key, const size_t step) { return key_switched(ct, rot_key, false, step); }

} // namespace ckks

// =====================================================================================================
// bgv.h
// =====================================================================================================
namespace bgv {

#ifndef HEHUB_AMD_BIND_REFERENCE   // bgv/arith.cpp:8-57 stay the reference's
BgvCt add(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    BgvCt r = ::hehub::add((const RlweCt &)a, (const RlweCt &)b);
    r.plain_modulus = a.plain_modulus;
    return r;
}

BgvCt sub(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    BgvCt r = ::hehub::sub((const RlweCt &)a, (const RlweCt &)b);
    r.plain_modulus = a.plain_modulus;
    return r;
}

// bgv/arith.cpp:17-57: the plaintext (one component modulo t) is lifted into the ciphertext's moduli, transformed
// and combined with the RLWE core operation
static RnsPolynomial lift_plain(const BgvCt &ct, const BgvPt &pt) {
    if (pt.component_count() != 1 || pt.modulus_at(0) != ct.plain_modulus) throw std::invalid_argument("plain moduli mismatch.");
    auto lifted = rns_base_transform(pt, ct[0].modulus_vec());
    ntt_negacyclic_inplace_lazy(lifted);
    return lifted;
}
BgvCt add_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::add_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
BgvCt sub_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::sub_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
BgvCt mult_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::mult_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
#endif

BgvQuadraticCt mult_low_level(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    auto q = mult_low_level_common<BgvQuadraticCt>(a, b);
    q.plain_modulus = a.plain_modulus;
    return q;
}

BgvCt relinearize(const BgvQuadraticCt &ct, const RlweKsk &key) {
    BgvCt r = relinearize_common(ct, key, true);
    r.plain_modulus = ct.plain_modulus;
    return r;
}

void mod_switch_inplace(BgvCt &ct, size_t dropping_primes) {   // mod_switch.cpp:80-90
    if (dropping_primes == 1) {
        drop_last_prime(ct, true, ct.plain_modulus);
    } else if (dropping_primes >= 2) {
        throw "under development";
    } else {
        throw std::invalid_argument("The number of primes to be dropped is not positive.");
    }
}

} // namespace bgv

} // namespace hehub
