// scheme_calls.cpp -- hehub's functions on the hot path, with hehub's names, argument meaning and exceptions:
// the operators of rns.h, mod_arith.h, ntt.h, permutation.h, rlwe.h / rgsw.h, ckks.h, bgv.h.  Each one checks what the reference
// checks, in its order, then runs (or records) one engine call.
#include "layer.hpp"

namespace hehub {

namespace detail {

// a result polynomial of the given shape (no host words are allocated for it in the own-mirror build)
RnsPolynomial result_poly(size_t n, size_t limbs, const std::vector<u64> &moduli, PolyRepForm form) {
    RnsPolynomial p;
    Access::shape(p, n, limbs, moduli);
    p.rep_form = form;
    return p;
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

#ifndef HEHUB_AMD_BIND_REFERENCE
std::unique_ptr<amd::PendingOp> new_op(amd::OpKind kind, size_t logn, size_t L, const std::vector<u64> &mod,
                                       std::initializer_list<const RnsIntVec *> operands, size_t in_limbs, size_t out_words);
#endif

void dev_binary(Bin op, size_t n, size_t L, const u64 *m, size_t batch, const u64 *a, const u64 *b, u64 *out) {
    auto *ctx = amd::cur();
    if (op == Bin::add) check(hp_dev_poly_add(ctx, n, L, m, batch, a, b, out));
    if (op == Bin::sub) check(hp_dev_poly_sub(ctx, n, L, m, batch, a, b, out));
    if (op == Bin::mul) check(hp_dev_poly_mul(ctx, n, L, m, batch, a, b, out));
}

// self (op)= b on the first L limbs, in place
void run_inplace(Bin op, RnsIntVec &self, const RnsIntVec &b, size_t L) {
    const size_t n = self.dimension();
    if (L == 0 || n == 0) return;
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (op != Bin::mul && n >= 2 && amd::deferred()) {   // recorded: self becomes the placeholder of the sum (the plaintext sums of a loop run as one batch)
        OpScope scope({Access::home(self), Access::home(b)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::PolyAddSub, lg, L, self.modulus_vec(), {&self, &b}, L, L * n);
        rec->sub = op == Bin::sub;
        Access::bind_block(self, amd::record(std::move(rec)), 0);
        return;
    }
#endif
    OpScope scope({Access::home(self), Access::home(b)});
    Src sb = Access::in(b, L);
#ifndef HEHUB_AMD_BIND_REFERENCE
    u64 *p = Access::inout(self);   // the vector's own device words: hp_dev_poly_* allow d_out == d_a
    dev_binary(op, n, L, self.modulus_vec().data(), 1, p, sb.p, p);
#else
    Src sa = Access::in(self, L);
    Dst d(L * n);
    dev_binary(op, n, L, self.modulus_vec().data(), 1, sa.p, sb.p, d.p);
    Access::bind(self, d, 0, L);
#endif
}

void scalar_mul(RnsIntVec &self, const std::vector<u64> &scalars) {
    const size_t n = self.dimension(), L = self.component_count();
    if (L == 0) return;
    OpScope op({Access::home(self)});
#ifndef HEHUB_AMD_BIND_REFERENCE
    u64 *p = Access::inout(self);
    check(hp_dev_poly_scalar_mul(amd::cur(), n, L, self.modulus_vec().data(), 1, scalars.data(), p, p));
#else
    Src s = Access::in(self, L);
    Dst d(L * n);
    check(hp_dev_poly_scalar_mul(amd::cur(), n, L, self.modulus_vec().data(), 1, scalars.data(), s.p, d.p));
    Access::bind(self, d, 0, L);
#endif
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

// Device copy of a key-switching key: u64[L][2][L+1][N], one block, assembled from the key's 2L polynomials.  A key is
// 2L(L+1) limbs (55 MiB at N=32768, L=10) and the same object call after call, so assembling it every time dominates.
//   own mirror:        the last HEHUB_AMD_KEY_CACHE (default 64) keys stay resident, recognised EXACTLY: every polynomial
//                      carries a stamp that changes whenever its words can have changed (RnsIntVec, hehub.hpp)
//   binding hehub's:   with HEHUB_AMD_KEY_CACHE=<entries> keys stay resident, recognised by the address of their first
//                      limb, their shape and four sampled words of every limb (a key that is modified in place between
//                      calls without touching any sampled word would go unnoticed: that is why the cache is opt-in there);
//                      default: the key is staged per call
DevKey::DevKey(const RgswCt &rgsw, size_t L, size_t n) {
#ifndef HEHUB_AMD_BIND_REFERENCE
    // (64: a rotation key set stays resident -- 3.4 GiB of 288 at the C3 shape)
    static const size_t cap = std::getenv("HEHUB_AMD_KEY_CACHE") ? (size_t)std::atoi(std::getenv("HEHUB_AMD_KEY_CACHE")) : 64;
#else
    static const size_t cap = std::getenv("HEHUB_AMD_KEY_CACHE") ? (size_t)std::atoi(std::getenv("HEHUB_AMD_KEY_CACHE")) : 0;
#endif
    const size_t words = L * 2 * (L + 1) * n;
    if (cap == 0) {
        own_ = amd::alloc_block(words);
        amd::track_write(*own_);
        assemble(own_->p, rgsw, L, n);
        return;
    }
    std::vector<u64> sig{(u64)L, (u64)n, (u64)amd::cur_rank()};   // (a key is assembled once per device rank that uses it)
    for (size_t j = 0; j < L; j++)
        for (size_t h = 0; h < 2; h++) {
#ifndef HEHUB_AMD_BIND_REFERENCE
            sig.push_back(Access::stamp(rgsw[j][h]));
#else
            sig.push_back((u64)(uintptr_t)rgsw[j][h][0].data());
            for (size_t k = 0; k <= L; k++) {
                const u64 *w = rgsw[j][h][(int)k].data();
                sig.insert(sig.end(), {w[0], w[n / 3], w[(2 * n) / 3], w[n - 1]});
            }
#endif
        }
    typedef std::vector<std::pair<std::vector<u64>, amd::BlockRef>> Cache;   // most recently used last
    static std::mutex &mu = *new std::mutex;   // (never destroyed, like the pool)
    static Cache &cache = *new Cache;
    std::lock_guard<std::mutex> lock(mu);
    for (size_t i = 0; i < cache.size(); i++)
        if (cache[i].first == sig) {
            auto hit = cache[i];
            cache.erase(cache.begin() + i);
            cache.push_back(hit);
            own_ = hit.second;
            amd::track_read(*own_);
            return;
        }
    own_ = amd::alloc_block(words);
    amd::track_write(*own_);
    assemble(own_->p, rgsw, L, n);
    if (cache.size() >= cap) cache.erase(cache.begin());
    cache.emplace_back(std::move(sig), own_);
}
void DevKey::assemble(u64 *dst, const RgswCt &rgsw, size_t L, size_t n) {
    for (size_t j = 0; j < L; j++)
        for (size_t h = 0; h < 2; h++) {
            u64 *row = dst + ((j * 2 + h) * (L + 1)) * n;
#ifndef HEHUB_AMD_BIND_REFERENCE
            Src s = Access::in(rgsw[j][h], L + 1);   // (a key polynomial that lives on the device is copied there)
            check(hp_dev_copy(amd::cur(), (L + 1) * n, s.p, row));
            Access::drop_device_copy(rgsw[j][h]);     // the assembled block is the key's device form: no second 55 MiB
#else
            amd::poly_copy_h2d(row, rgsw[j][h], L + 1, n);
#endif
        }
}


#ifndef HEHUB_AMD_BIND_REFERENCE
// deferred mode: a recorded call over the given operand polynomials (host words are uploaded now, placeholders stay placeholders)
std::unique_ptr<amd::PendingOp> new_op(amd::OpKind kind, size_t logn, size_t L, const std::vector<u64> &mod,
                                       std::initializer_list<const RnsIntVec *> operands, size_t in_limbs, size_t out_words) {
    std::unique_ptr<amd::PendingOp> op(new amd::PendingOp);
    op->kind = kind; op->logn = logn; op->L = L; op->L0 = L; op->mod = mod; op->in_limbs = in_limbs; op->out_words = out_words;
    op->rank = amd::rank_of(amd::lane_set().cur);   // (the recording scope chose the rank from the operands: RecordScope)
    for (const RnsIntVec *v : operands) op->in.push_back(Access::ref(*v));
    return op;
}
// the polynomials of a recorded call's result: views [h * limbs * N, ..) of its placeholder
template <class Polys> void bind_placeholder(Polys &polys, size_t count, const amd::BlockRef &ph, size_t limbs, size_t n) {
    for (size_t h = 0; h < count; h++) Access::bind_block(polys[h], ph, h * limbs * n);
}
#endif

// the two polynomials of a result ciphertext as views of the block an engine call filled: [2][L][N]
RlweCt make_ct(size_t n, size_t L, const std::vector<u64> &moduli, const Dst &d) {
    RlweCt ct{result_poly(n, L, moduli, PolyRepForm::value), result_poly(n, L, moduli, PolyRepForm::value)};
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 2; h++) Access::bind_enqueue(ct[h], d, (size_t)h * L * n, L);
    amd::limb_copies_wait();
    for (int h = 0; h < 2; h++) Access::bind_finish(ct[h], d, (size_t)h * L * n, L);
#else
    for (int h = 0; h < 2; h++) Access::bind(ct[h], d, (size_t)h * L * n, L);
#endif
    return ct;
}

// shared body of ckks::relinearize / bgv::relinearize
RlweCt relinearize_common(const std::array<RnsPolynomial, 3> &quad, const RlweKsk &key, bool bgv) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(quad[2], key, mext);
    const size_t n = quad[2].dimension(), L = quad[2].component_count();
    if (bgv && L0 != L) throw std::invalid_argument("Inconsistent RGSW ciphertext.");   // no higher-level keys for the BGV quirk path
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(quad[0]), Access::home(quad[1]), Access::home(quad[2])}, 0);
        DevKey dk(key, L0, n);
        auto rec = new_op(amd::OpKind::Relin, quad[2].log_dimension(), L, mext, {&quad[0], &quad[1], &quad[2]}, L, 2 * L * n);
        rec->L0 = L0; rec->bgv = bgv; rec->key = dk.block();
        std::vector<u64> q(mext.begin(), mext.begin() + L);
        RlweCt ct{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)};
        bind_placeholder(ct, 2, amd::record(std::move(rec)), L, n);
        return ct;
    }
#endif
    OpScope op({Access::home(quad[0]), Access::home(quad[1]), Access::home(quad[2])});
    DevKey dk(key, L0, n);
    Src dq = amd::gather({&quad[0], &quad[1], &quad[2]}, L);
    Dst dout(2 * L * n);
    const size_t logn = quad[2].log_dimension();
    if (bgv) check(hp_dev_bgv_relinearize(amd::cur(), logn, L, mext.data(), 1 /* bgv.h:32 */, 1, dq.p, dk.p(), dout.p));
    else check(hp_dev_ckks_relinearize_at(amd::cur(), logn, L, L0, mext.data(), 1, dq.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    return make_ct(n, L, q, dout);
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
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct1[0]), Access::home(ct1[1]), Access::home(ct2[0]), Access::home(ct2[1])}, 0);
        auto rec = new_op(amd::OpKind::MultLow, ct1[0].log_dimension(), L, m1, {&ct1[0], &ct1[1], &ct2[0], &ct2[1]}, L, 3 * L * n);
        Quad quad;
        for (int h = 0; h < 3; h++) quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        bind_placeholder(quad, 3, amd::record(std::move(rec)), L, n);
        return quad;
    }
#endif
    // (a ciphertext with more limbs than L does not lie as [2][L][N]: gather() then copies the first L limbs of each half)
    OpScope op({Access::home(ct1[0]), Access::home(ct1[1]), Access::home(ct2[0]), Access::home(ct2[1])});
    Src d1 = amd::gather({&ct1[0], &ct1[1]}, L);
    Src d2 = amd::gather({&ct2[0], &ct2[1]}, L);
    Dst dq(3 * L * n);
    check(hp_dev_mult_low_level(amd::cur(), ct1[0].log_dimension(), L, m1.data(), 1, d1.p, d2.p, dq.p));
    Quad quad;
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 3; h++) {
        quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        Access::bind_enqueue(quad[h], dq, (size_t)h * L * n, L);
    }
    amd::limb_copies_wait();
    for (int h = 0; h < 3; h++) Access::bind_finish(quad[h], dq, (size_t)h * L * n, L);
#else
    for (int h = 0; h < 3; h++) {
        quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        Access::bind(quad[h], dq, (size_t)h * L * n, L);
    }
#endif
    return quad;
}

void drop_last_prime(RlweCt &ct, bool bgv, u64 t) {
    check_ct_wellformed(ct);
    const size_t n = ct[0].dimension(), L = ct[0].component_count(), logn = ct[0].log_dimension();
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct[0]), Access::home(ct[1])}, 0);
        auto rec = new_op(amd::OpKind::Drop, logn, L, ct[0].modulus_vec(), {&ct[0], &ct[1]}, L, 2 * (L - 1) * n);
        rec->bgv = bgv; rec->t = t;
        const amd::BlockRef ph = amd::record(std::move(rec));
        for (int h = 0; h < 2; h++) {
            ct[h].remove_components();
            Access::bind_block(ct[h], ph, (size_t)h * (L - 1) * n);
        }
        return;
    }
#endif
    OpScope op({Access::home(ct[0]), Access::home(ct[1])});
    Src din = amd::gather({&ct[0], &ct[1]}, L);
    Dst dout(2 * (L - 1) * n);
    const std::vector<u64> m(ct[0].modulus_vec());
    if (bgv) check(hp_dev_bgv_mod_switch(amd::cur(), logn, L, m.data(), t, 1, din.p, dout.p));
    else check(hp_dev_ckks_rescale(amd::cur(), logn, L, m.data(), 1, din.p, dout.p));
#ifdef HEHUB_AMD_BIND_REFERENCE
    // remove_components hands the last limb's block back to hehub's pool, whose free list writes its link into the block's first
    // word (allocator.h:67-71): the (asynchronous) upload of that limb must have happened by then
    amd::limb_copies_wait();
#endif
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 2; h++) {
        ct[h].remove_components();
        Access::bind_enqueue(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
    }
    amd::limb_copies_wait();
    for (int h = 0; h < 2; h++) Access::bind_finish(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
#else
    for (int h = 0; h < 2; h++) {
        ct[h].remove_components();
        Access::bind(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
    }
#endif
}

} // namespace detail

using namespace detail;


// =====================================================================================================
// rns.h / rns.cpp: operators
// =====================================================================================================
const RnsIntVec &operator+=(RnsIntVec &self, const RnsIntVec &b) {
    run_inplace(Bin::add, self, b, check_addsub(self, b));
    return self;
}

const RnsIntVec &operator-=(RnsIntVec &self, const RnsIntVec &b) {
    run_inplace(Bin::sub, self, b, check_addsub(self, b));
    return self;
}

RnsIntVec operator*(const RnsIntVec &a, const RnsIntVec &b) {
    if (a.dimension() != b.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    auto components = std::min(a.component_count(), b.component_count());
    auto moduli(a.modulus_vec()), b_moduli(b.modulus_vec());
    moduli.resize(components);
    b_moduli.resize(components);
    if (moduli != b_moduli) throw std::invalid_argument("Operands' moduli mismatch.");
    RnsIntVec result;
    Access::shape(result, a.dimension(), components, moduli);
    const size_t n = a.dimension();
    if (components == 0 || n == 0) return result;
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred() && n >= 2) {
        OpScope scope({Access::home(a), Access::home(b)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::PolyMul, lg, components, moduli, {&a, &b}, components, components * n);
        Access::bind_block(result, amd::record(std::move(rec)), 0);
        return result;
    }
#endif
    OpScope op({Access::home(a), Access::home(b)});
    Src sa = Access::in(a, components), sb = Access::in(b, components);
    Dst d(components * n);
    dev_binary(Bin::mul, n, components, moduli.data(), 1, sa.p, sb.p, d.p);
    Access::bind(result, d, 0, components);
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
void batched_barrett_lazy(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett_lazy(amd::cur(), q, n, v)); }
#ifndef HEHUB_AMD_BIND_REFERENCE   // inline in the reference's mod_arith.h:18-25,58-63
void batched_barrett(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett(amd::cur(), q, n, v)); }
void batched_reduce_strict(const u64 q, const size_t n, u64 v[]) { check(hp_batched_reduce_strict(amd::cur(), q, n, v)); }
#endif
void batched_mul_mod_hybrid_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_hybrid_lazy(amd::cur(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_hybrid(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_hybrid_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_mul_mod_barrett_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_barrett_lazy(amd::cur(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_barrett(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_barrett_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_montgomery_128_lazy(const u64 q, const size_t len, const u128 in[], u64 out[]) {
    check(hp_batched_montgomery_128_lazy(amd::cur(), q, len, reinterpret_cast<const u64 *>(in), out));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // mod_arith.h:65-72 (inline) and mod_arith.cpp:136-149 stay the reference's
void reduce_strict(RnsPolynomial &p) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L == 0) return;
    OpScope op({Access::home(p)});
    check(hp_dev_poly_reduce_strict(amd::cur(), n, L, p.modulus_vec().data(), 1, Access::inout(p)));
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
    check(hp_ntt_negacyclic_inplace_lazy(amd::cur(), logn, q, c));
}
void intt_negacyclic_inplace_lazy(const size_t logn, const u64 q, u64 v[]) {
    check(hp_intt_negacyclic_inplace_lazy(amd::cur(), logn, q, v));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // ntt.h:41-92 (inline per-limb loops) stay the reference's
static void poly_transform(RnsPolynomial &p, bool inverse, bool strict) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L && n >= 2 && amd::deferred()) {   // recorded: the transforms of a loop's plaintexts run as one batch (conj = inverse, sub = strict)
        OpScope scope({Access::home(p)}, 0);
        auto rec = new_op(amd::OpKind::Transform, p.log_dimension(), L, p.modulus_vec(), {&p}, L, L * n);
        rec->conj = inverse; rec->sub = strict;
        Access::bind_block(p, amd::record(std::move(rec)), 0);
    } else if (L) {
        OpScope op({Access::home(p)});
        u64 *d = Access::inout(p);
        if (inverse) check(hp_dev_intt(amd::cur(), p.log_dimension(), L, p.modulus_vec().data(), 1, d, strict ? 1 : 0));
        else check(hp_dev_ntt(amd::cur(), p.log_dimension(), L, p.modulus_vec().data(), 1, d));
    }
    p.rep_form = inverse ? PolyRepForm::coeff : PolyRepForm::value;
}
void ntt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, false, false); }
void intt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, true, false); }
void intt_negacyclic_inplace(RnsPolynomial &p) { poly_transform(p, true, true); }
#endif

void cache_ntt_factors_strict(const u64 logn, const std::vector<u64> &moduli) {
    check(hp_cache_ntt_factors_strict(amd::cur(), logn, moduli.data(), moduli.size()));
}

// =====================================================================================================
// permutation.h
// =====================================================================================================
static RnsPolynomial gather(const RnsPolynomial &p, bool is_cycle, size_t step) {
    if (p.rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    const size_t n = p.dimension(), L = p.component_count();
    RnsPolynomial out = result_poly(n, L, p.modulus_vec(), PolyRepForm::value);
    if (L == 0) return out;
    OpScope op({Access::home(p)});
    Src din = Access::in(p, L);
    Dst dout(L * n);
    if (is_cycle) check(hp_dev_poly_cycle(amd::cur(), p.log_dimension(), L, 1, step, din.p, dout.p));
    else check(hp_dev_poly_involution(amd::cur(), p.log_dimension(), L, 1, din.p, dout.p));
    Access::bind(out, dout, 0, L);
    return out;
}
RnsPolynomial cycle(const RnsPolynomial &p, const size_t step) { return gather(p, true, step); }
RnsPolynomial involution(const RnsPolynomial &p) { return gather(p, false, 0); }

// =====================================================================================================
// rlwe.h / rgsw.h
// =====================================================================================================
#ifndef HEHUB_AMD_BIND_REFERENCE   // rlwe.cpp:83-101: thin compositions of the operators above
// both halves in ONE launch when the two ciphertexts have the same shape (the result's halves then lie side by side, ready
// for the next scheme-level call); the checks are operator+= 's (rns.h:207-218, rns.cpp:59-72), half by half, in its order
static RlweCt addsub(const RlweCt &a, const RlweCt &b, bool sub) {
    size_t L[2];
    for (int h = 0; h < 2; h++) {
        if (a[h].rep_form != b[h].rep_form) throw std::invalid_argument("Operands are in different representation form.");
        L[h] = check_addsub(a[h], b[h]);
    }
    const size_t n = a[0].dimension();
    const bool same = L[0] == L[1] && L[0] > 0 && a[1].dimension() == n && a[0].modulus_vec() == a[1].modulus_vec() &&
                      b[0].component_count() == L[0] && b[1].component_count() == L[0];
    if (same && amd::deferred()) {
        OpScope scope({Access::home(a[0]), Access::home(a[1]), Access::home(b[0]), Access::home(b[1])}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::AddSub, lg, L[0], a[0].modulus_vec(), {&a[0], &a[1], &b[0], &b[1]}, L[0], 2 * L[0] * n);
        rec->sub = sub;
        RlweCt r{result_poly(n, L[0], a[0].modulus_vec(), a[0].rep_form), result_poly(n, L[0], a[1].modulus_vec(), a[1].rep_form)};
        bind_placeholder(r, 2, amd::record(std::move(rec)), L[0], n);
        return r;
    }
    OpScope op({Access::home(a[0]), Access::home(a[1]), Access::home(b[0]), Access::home(b[1])});
    if (!same) return sub ? RlweCt{a[0] - b[0], a[1] - b[1]} : RlweCt{a[0] + b[0], a[1] + b[1]};
    Src sa = amd::gather({&a[0], &a[1]}, L[0]), sb = amd::gather({&b[0], &b[1]}, L[0]);
    Dst d(2 * L[0] * n);
    dev_binary(sub ? Bin::sub : Bin::add, n, L[0], a[0].modulus_vec().data(), 2, sa.p, sb.p, d.p);
    RlweCt r{result_poly(n, L[0], a[0].modulus_vec(), a[0].rep_form), result_poly(n, L[0], a[1].modulus_vec(), a[1].rep_form)};
    for (int h = 0; h < 2; h++) Access::bind(r[h], d, (size_t)h * L[0] * n, L[0]);
    return r;
}
RlweCt add(const RlweCt &a, const RlweCt &b) { return addsub(a, b, false); }
RlweCt sub(const RlweCt &a, const RlweCt &b) { return addsub(a, b, true); }
RlweCt add_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] + pt, ct[1]}; }
RlweCt sub_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] - pt, ct[1]}; }
RlweCt mult_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] * pt, ct[1] * pt}; }
#endif

RlweCt ext_prod_montgomery(const RlwePt &pt, const RgswCt &rgsw) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(pt, rgsw, mext);
    const size_t n = pt.dimension(), L = pt.component_count();
    OpScope op({Access::home(pt)});
    DevKey dk(rgsw, L0, n);
    Src dp = Access::in(pt, L);
    Dst dout(2 * (L + 1) * n);
    check(hp_dev_ext_prod_montgomery_at(amd::cur(), pt.log_dimension(), L, L0, mext.data(), 1, dp.p, dk.p(), dout.p));
    return make_ct(n, L + 1, mext, dout);
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
    RnsPolynomial pt = result_poly(n, L, m1, PolyRepForm::coeff);
    if (L == 0) return pt;
    OpScope op({Access::home(c0), Access::home(c1)});
    Src dct = amd::gather({&c0, &c1}, L), dsk = Access::in(sk, L);
    Dst dpt(L * n);
    check(hp_dev_rlwe_decrypt_core(amd::cur(), c0.log_dimension(), L, m1.data(), 1, dct.p, dsk.p, dpt.p));
    Access::bind(pt, dpt, 0, L);
    return pt;
}

#ifndef HEHUB_AMD_BIND_REFERENCE
// rns_transform.cpp:106-127 on the device: one -> many (:11-37) and many -> one (:39-104, both branches)
RnsPolynomial rns_base_transform(RnsPolynomial in, const std::vector<u64> &new_moduli) {
    if (in.rep_form == PolyRepForm::value)
        throw std::logic_error("Trying to perform RNS base transformation on NTT values.");
    const size_t n = in.dimension(), L = in.component_count();
    // recorded in deferred mode: the plaintext lifts of a loop of bgv::add_plain / sub_plain / mult_plain (bgv/arith.cpp:17-57) run as
    // one batch.  (one -> many only: many -> one is decrypt's, whose caller looks at the words next, and its preconditions -- odd,
    // pairwise coprime moduli -- are reported by the call itself)
    if (amd::deferred() && n >= 2 && L == 1 && !new_moduli.empty() && in.modulus_at(0) >= 2) {
        OpScope scope({Access::home(in)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        RnsPolynomial out = result_poly(n, new_moduli.size(), new_moduli, PolyRepForm::coeff);
        auto rec = new_op(amd::OpKind::BaseConv, lg, new_moduli.size(), new_moduli, {&in}, 1, new_moduli.size() * n);
        rec->t = in.modulus_at(0);
        Access::bind_block(out, amd::record(std::move(rec)), 0);
        return out;
    }
    OpScope op({Access::home(in)});
    if (L == 1) {
        RnsPolynomial out = result_poly(n, new_moduli.size(), new_moduli, PolyRepForm::coeff);
        if (new_moduli.empty()) return out;
        Src din = Access::in(in, 1);
        Dst dout(new_moduli.size() * n);
        check(hp_dev_rns_base_from_single(amd::cur(), n, in.modulus_at(0), new_moduli.size(), new_moduli.data(), 1, din.p, dout.p));
        Access::bind(out, dout, 0, new_moduli.size());
        return out;
    }
    if (new_moduli.size() == 1) {   // both branches of rns_transform.cpp:39-104 on the device
        RnsPolynomial out = result_poly(n, 1, new_moduli, PolyRepForm::coeff);
        Src din = Access::in(in, L);
        Dst dout(n);
        check(hp_dev_rns_base_to_single(amd::cur(), n, L, in.modulus_vec().data(), new_moduli[0], 1, din.p, dout.p));
        Access::bind(out, dout, 0, 1);
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
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct[0]), Access::home(ct[1])}, 0);
        DevKey dk(key, L0, n);
        auto rec = new_op(amd::OpKind::KeySwitch, logn, L, mext, {&ct[0], &ct[1]}, L, 2 * L * n);
        rec->L0 = L0; rec->conj = conj; rec->step = conj ? 0 : step; rec->key = dk.block();
        std::vector<u64> q(mext.begin(), mext.begin() + L);
        CkksCt r(RlweCt{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)});
        bind_placeholder(r, 2, amd::record(std::move(rec)), L, n);
        r.scaling_factor = ct.scaling_factor;
        return r;
    }
#endif
    OpScope op({Access::home(ct[0]), Access::home(ct[1])});
    DevKey dk(key, L0, n);
    Src dct = amd::gather({&ct[0], &ct[1]}, L);
    Dst dout(2 * L * n);
    if (conj) check(hp_dev_ckks_conjugate_at(amd::cur(), logn, L, L0, mext.data(), 1, dct.p, dk.p(), dout.p));
    else check(hp_dev_ckks_rotate_at(amd::cur(), logn, L, L0, mext.data(), 1, step, dct.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    CkksCt r = make_ct(n, L, q, dout);
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
