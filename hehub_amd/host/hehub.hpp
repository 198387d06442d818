// hehub.hpp -- C++ host layer that mirrors primihub/hehub's public interface for the ring-arithmetic
// hot path and forwards every operation to the HIP engine through the C ABI (include/hehub_amd.h).
//
// Same names, argument meaning and error behaviour as the reference headers it stands in for
// (citations are relative to the hehub source tree):
//   RnsIntVec / RnsPolynomial and operators     src/fhe/common/rns.h:15-280, rns.cpp:58-171
//   ntt / intt / reduce_strict / batched_*       src/fhe/common/ntt.h:33-102, mod_arith.h:16-78
//   cycle / involution                           src/fhe/common/permutation.h:90-92
//   RlweCt, add/sub/mult_plain_core              src/fhe/primitives/rlwe.h:27,98-134
//   RgswCt, RlweKsk, ext_prod_montgomery         src/fhe/primitives/rgsw.h:20,51, keys.h:19-32
//   ckks::{add,sub,mult_low_level,relinearize,mult,rescale_inplace}   src/fhe/ckks/ckks.h:73-313
//   bgv::{add,sub,mult_low_level,relinearize,mod_switch_inplace}      src/fhe/bgv/bgv.h:24-167
//
// What is NOT here (out of scope): encoders, samplers, key generation, BigInt,
// rns_base_transform.  A hehub build keeps its own files for those and links this layer for the rest
// (INTEGRATION.md).
//
// Operands stay in HBM between calls (see RnsIntVec below).  Independent single calls overlap on the device (lanes), and
// hehub_amd_ext.hpp (included at the end) adds batched forms of the scheme-level calls: std::vector<CkksCt> in, ONE engine call.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

struct hp_ctx;

namespace hehub {

using u64 = uint64_t;
using u128 = unsigned __int128;

// Stand-alone mirror of hehub's RNS vector (rns.h:15-115): the public names a caller of hehub uses.  Only built when hehub's own
// headers are not (the real binding compiles against those: -DHEHUB_AMD_BIND_REFERENCE).
//
// DEVICE RESIDENCY.  hehub's limbs are pooled host blocks (allocator.h:105-220); here a vector's limbs live where they were
// last written: results of engine calls stay in HBM (one contiguous [L][N] view of a pooled device block) and are copied to the
// host only when somebody looks at the words -- operator[], components(), begin() / end() / last(), operator== -- while words
// written on the host are uploaded by the first engine call that consumes them.  Two flags per vector say which copy is
// current; a non-const host accessor hands out writable words, so it marks the device copy stale.  A program that keeps to
// hehub's API between encrypt and decrypt (ckks::mult / add / rotate / rescale_inplace ... chains) therefore crosses PCIe
// once per input ciphertext and once per result it actually reads.  Value semantics are hehub's: a copy is a deep copy
// (device to device when the device copy is current), a move leaves the source empty (allocator.h:113-155).  Like hehub
// itself (process-global unsynchronised caches and pools) the layer is for ONE thread at a time.
namespace amd {
struct DevBlock;   // a pooled device allocation (layer.hpp, block_pool.cpp)
struct Access;     // the layer's view of a vector's two copies (residency.hpp)
} // namespace amd

class RnsIntVec {
public:
    using ComponentData = std::vector<u64>;   // one limb: N words modulo modulus_at(k)
    struct Params {
        size_t dimension = 0;
        size_t component_count;
        std::vector<u64> moduli;
    };

    RnsIntVec() = default;
    RnsIntVec(size_t dimension, size_t components, const std::vector<u64> &moduli);
    RnsIntVec(const Params &params);
    RnsIntVec(const RnsIntVec &o);
    RnsIntVec(RnsIntVec &&o) noexcept;
    RnsIntVec &operator=(const RnsIntVec &o);
    RnsIntVec &operator=(RnsIntVec &&o) noexcept;
    ~RnsIntVec() = default;

    // shape
    size_t log_dimension() const { return logn_; }
    size_t dimension() const { return count_ == 0 && !logn_ ? 0 : (size_t)1 << logn_; }
    size_t component_count() const { return count_; }
    Params params() const { return Params{dimension(), count_, q_}; }
    u64 modulus_at(int k) const { return q_[k]; }
    const std::vector<u64> &modulus_vec() const { return q_; }
    bool operator==(const RnsIntVec &o) const;

    // limbs (host words: these synchronise with the device copy, see above).  A reference obtained here shows the vector's
    // words as of that moment: after the next engine call on the vector, ask again.
    ComponentData &operator[](int k) { return host_rw()[k]; }
    const ComponentData &operator[](int k) const { return host_ro_limb(k); }
    /// read-only words of limb k / of all limbs WITHOUT giving up the device copy, also on a non-const vector (`ct[0].view(k)[i]`):
    /// what a caller that only looks at a result should use -- operator[] on a non-const vector must assume a write
    /// (a look at ONE limb of a device-resident vector downloads that limb, not the vector)
    const ComponentData &view(int k) const { return host_ro_limb(k); }
    const std::vector<ComponentData> &view() const { return host_ro(); }
    std::vector<ComponentData> &components() { return host_rw(); }
    const std::vector<ComponentData> &components() const { return host_ro(); }
    auto begin() { return host_rw().begin(); }
    auto end() { return host_rw().end(); }
    auto last() { return host_rw().end() - 1; }
    auto begin() const { return host_ro().cbegin(); }
    auto end() const { return host_ro().cend(); }
    auto last() const { return host_ro().cend() - 1; }

    void add_components(const std::vector<u64> &new_moduli, size_t adding = 1);
    void remove_components(size_t removing = 1);

    /// true when the current words are in HBM only (no host copy has been made since the last engine call wrote them)
    bool device_resident() const { return dev_ok_ && !host_ok_; }

private:
    friend struct amd::Access;
    std::vector<ComponentData> &host_rw();               // current host words, writable: the device copy becomes stale
    const std::vector<ComponentData> &host_ro() const;   // current host words, read-only: both copies stay current
    const ComponentData &host_ro_limb(int k) const;      // ... of limb k alone (the other limbs' host words may stay stale until asked for)
    size_t logn_ = 0, count_ = 0;
    std::vector<u64> q_;
    mutable std::vector<ComponentData> limbs_;       // host copy, count_ limbs when host_ok_
    mutable std::shared_ptr<amd::DevBlock> blk_;     // device copy: limb k at block + off_ + k * N words
    mutable size_t off_ = 0;
    mutable bool host_ok_ = true, dev_ok_ = false;
    mutable unsigned long long stamp_ = 0;           // changes whenever the words may have changed (key cache, scheme_calls.cpp)
    mutable unsigned long long limb_mask_ = 0, mask_stamp_ = 0;   // while !host_ok_: limbs downloaded one by one since the words last changed (valid for stamp_ == mask_stamp_)
};

class RnsPolynomial : public RnsIntVec {
public:
    using RnsIntVec::RnsIntVec;
    enum class RepForm { coeff, value };
    RnsPolynomial() {}
    RnsPolynomial(RnsIntVec &&v) : RnsIntVec(std::move(v)) {}
    RepForm rep_form = RepForm::coeff;
};
using RnsPolyParams = RnsPolynomial::Params;
using PolyRepForm = RnsPolynomial::RepForm;

// ---- rns.h operators ---------------------------------------------------------------------------
const RnsIntVec &operator+=(RnsIntVec &self, const RnsIntVec &b);
const RnsIntVec &operator-=(RnsIntVec &self, const RnsIntVec &b);
RnsIntVec operator*(const RnsIntVec &a, const RnsIntVec &b);
const RnsIntVec &operator*=(RnsIntVec &self, const u64 small_scalar);
const RnsIntVec &operator*=(RnsIntVec &self, const std::vector<u64> &rns_scalar);
inline RnsIntVec operator+(const RnsIntVec &a, const RnsIntVec &b) { auto r(a); r += b; return r; }
inline RnsIntVec operator-(const RnsIntVec &a, const RnsIntVec &b) { auto r(a); r -= b; return r; }

const RnsPolynomial &operator+=(RnsPolynomial &self, const RnsPolynomial &b);
const RnsPolynomial &operator-=(RnsPolynomial &self, const RnsPolynomial &b);
RnsPolynomial operator*(const RnsPolynomial &a, const RnsPolynomial &b);
inline RnsPolynomial operator+(const RnsPolynomial &a, const RnsPolynomial &b) { auto r(a); r += b; return r; }
inline RnsPolynomial operator-(const RnsPolynomial &a, const RnsPolynomial &b) { auto r(a); r -= b; return r; }
inline const RnsPolynomial &operator*=(RnsPolynomial &self, const RnsPolynomial &b) { auto t(self); return self = t * b; }
const RnsPolynomial &operator*=(RnsPolynomial &self, const u64 small_scalar);
const RnsPolynomial &operator*=(RnsPolynomial &self, const std::vector<u64> &rns_scalar);
inline RnsPolynomial operator*(const RnsPolynomial &p, const std::vector<u64> &s) { auto c(p); c *= s; return c; }

// ---- mod_arith.h ---------------------------------------------------------------------------------
void batched_barrett_lazy(const u64 modulus, const size_t vec_len, u64 vec[]);
void batched_barrett(const u64 modulus, const size_t vec_len, u64 vec[]);
void batched_mul_mod_hybrid_lazy(const u64 modulus, const size_t vec_len, const u64 in_vec1[], const u64 in_vec2[],
                                 u64 out_vec[]);
void batched_mul_mod_hybrid(const u64 modulus, const size_t vec_len, const u64 in_vec1[], const u64 in_vec2[],
                            u64 out_vec[]);
void batched_mul_mod_barrett_lazy(const u64 modulus, const size_t vec_len, const u64 in_vec1[], const u64 in_vec2[],
                                  u64 out_vec[]);
void batched_mul_mod_barrett(const u64 modulus, const size_t vec_len, const u64 in_vec1[], const u64 in_vec2[],
                             u64 out_vec[]);
void batched_montgomery_128_lazy(const u64 modulus, const size_t len, const u128 in[], u64 out[]);
void batched_reduce_strict(const u64 modulus, const size_t vec_len, u64 vec[]);
void reduce_strict(RnsPolynomial &rns_poly);
/// scalar helper kept on the host exactly as in the reference (mod_arith.h:74-78)
inline u64 mul_mod_harvey_lazy(const u64 modulus, const u64 in1, const u64 in2, const u64 in2_harvey) {
    u64 approx_quotient = (u128)in1 * in2_harvey >> 64;
    return (u128)in1 * in2 - (u128)approx_quotient * modulus;
}
u64 inverse_mod_prime(const u64 elem, const u64 prime);

// ---- ntt.h ------------------------------------------------------------------------------------------
void ntt_negacyclic_inplace_lazy(const size_t log_dimension, const u64 modulus, u64 coeffs[]);
void ntt_negacyclic_inplace_lazy(RnsPolynomial &rns_poly);
void intt_negacyclic_inplace_lazy(const size_t log_dimension, const u64 modulus, u64 values[]);
void intt_negacyclic_inplace_lazy(RnsPolynomial &rns_poly);
void intt_negacyclic_inplace(RnsPolynomial &rns_poly);
void cache_ntt_factors_strict(const u64 log_dimension, const std::vector<u64> &moduli);

// ---- permutation.h -------------------------------------------------------------------------------------
RnsPolynomial cycle(const RnsPolynomial &poly_ntt, const size_t step);
RnsPolynomial involution(const RnsPolynomial &poly_ntt);

// ---- rlwe.h / rgsw.h / keys.h ---------------------------------------------------------------------------
using RlwePt = RnsPolynomial;
using RlweCt = std::array<RnsPolynomial, 2>;
RlweCt add(const RlweCt &ct1, const RlweCt &ct2);
RlweCt sub(const RlweCt &ct1, const RlweCt &ct2);
RlweCt add_plain_core(const RlweCt &ct, const RlwePt &pt);
RlweCt sub_plain_core(const RlweCt &ct, const RlwePt &pt);
RlweCt mult_plain_core(const RlweCt &ct, const RlwePt &pt);
struct RlweSk : public RnsPolynomial {   // rlwe.h:34 (sampling the key is the caller's business here)
    using RnsPolynomial::RnsPolynomial;
    RlweSk() {}
    RlweSk(RnsPolynomial &&p) : RnsPolynomial(std::move(p)) {}
};
RlwePt decrypt_core(const RlweCt &ct, const RlweSk &sk);   // rlwe.cpp:74-81
// rns_transform.h: one modulus -> many, and many -> one (small-coefficient and CRT branches, rns_transform.cpp:39-104)
RnsPolynomial rns_base_transform(RnsPolynomial input_rns_poly, const std::vector<u64> &new_moduli);

using RgswCt = std::vector<RlweCt>;
struct RlweKsk : public RgswCt {
    using RgswCt::RgswCt;
    RlweKsk() {}
    RlweKsk(RgswCt &&rgsw) : RgswCt(std::move(rgsw)) {}
};
struct RotKey : public RlweKsk {   // keys.h:63-68 (key generation itself -- sampling -- stays with the caller)
    using RlweKsk::RlweKsk;
    RotKey() {}
    RotKey(RlweKsk &&ksk) : RlweKsk(std::move(ksk)) {}
    size_t step = 0;
};
RlweCt ext_prod_montgomery(const RlwePt &pt, const RgswCt &rgsw);

// ---- ckks.h -----------------------------------------------------------------------------------------------
namespace ckks {
struct CkksCt : public RlweCt {
    using RlweCt::RlweCt;
    CkksCt() {}
    CkksCt(RlweCt &&other) : RlweCt(std::move(other)) {}
    double scaling_factor = 1.0;
};
struct CkksQuadraticCt : public std::array<RnsPolynomial, 3> {
    double scaling_factor = 1.0;
};
struct CkksPt : public RlwePt {   // ckks.h:58-67
    using RlwePt::RlwePt;
    CkksPt() {}
    CkksPt(RlwePt &&other) : RlwePt(std::move(other)) {}
    double scaling_factor = 1.0;
};
CkksCt add(const CkksCt &ct1, const CkksCt &ct2);
CkksCt sub(const CkksCt &ct1, const CkksCt &ct2);
CkksCt add_plain(const CkksCt &ct, const CkksPt &pt);
CkksCt sub_plain(const CkksCt &ct, const CkksPt &pt);
CkksCt mult_plain(const CkksCt &ct, const CkksPt &pt);
CkksQuadraticCt mult_low_level(const CkksCt &ct1, const CkksCt &ct2);
CkksCt relinearize(const CkksQuadraticCt &ct, const RlweKsk &relin_key);
inline CkksCt mult(const CkksCt &ct1, const CkksCt &ct2, const RlweKsk &relin_key) {
    auto ct_prod = mult_low_level(ct1, ct2);
    return relinearize(ct_prod, relin_key);
}
CkksCt conjugate(const CkksCt &ct, const RlweKsk &conj_key);
CkksCt rotate(const CkksCt &ct, const RlweKsk &rot_key, const size_t step);
inline CkksCt rotate(const CkksCt &ct, const RotKey &rot_key) { return rotate(ct, rot_key, rot_key.step); }   // ckks.h:303
void rescale_inplace(CkksCt &ct, size_t dropping_primes = 1);
} // namespace ckks

// ---- bgv.h --------------------------------------------------------------------------------------------------
namespace bgv {
struct BgvCt : public RlweCt {
    using RlweCt::RlweCt;
    BgvCt() {}
    BgvCt(RlweCt &&other) : RlweCt(std::move(other)) {}
    u64 plain_modulus = 1;
};
struct BgvQuadraticCt : public std::array<RnsPolynomial, 3> {
    u64 plain_modulus = 1;
};
using BgvPt = RlwePt;   // bgv.h:18: one component modulo the plain modulus
BgvCt add(const BgvCt &ct1, const BgvCt &ct2);
BgvCt sub(const BgvCt &ct1, const BgvCt &ct2);
BgvCt add_plain(const BgvCt &ct, const BgvPt &pt);
BgvCt sub_plain(const BgvCt &ct, const BgvPt &pt);
BgvCt mult_plain(const BgvCt &ct, const BgvPt &pt);
BgvQuadraticCt mult_low_level(const BgvCt &ct1, const BgvCt &ct2);
BgvCt relinearize(const BgvQuadraticCt &ct, const RlweKsk &relin_key);
void mod_switch_inplace(BgvCt &ct, size_t dropping_primes = 1);
} // namespace bgv

using CkksCt = ckks::CkksCt;
using BgvCt = bgv::BgvCt;

} // namespace hehub

// the engine handle, parity level, PCIe accounting, lanes and the BATCHED forms of the scheme-level calls (namespace hehub::amd)
#include "hehub_amd_ext.hpp"
