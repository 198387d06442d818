// hp_tables.h -- host-side construction of per-modulus constants and twiddle
// tables for the HIP kernels.  Values follow hehub's NTTFactors
// (src/fhe/common/ntt.cpp:41-105) and mod_arith.cpp:49-62; the *layouts* are
// the engine's own (reference order for the simple kernels, kernel order for
// the register/LDS-tiled ones).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hp {

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct Pair {
    u64 w, wh;
};

struct ModConsts {
    u64 q, two_q, neg_q, mqinv, r64, r64h, barrett_c;
    uint32_t k, fix;
};

// scalar helpers (host)
u64 harvey_quotient(u64 b, u64 q);                 // floor(b * 2^64 / q)
u64 mul_mod(u64 a, u64 b, u64 q);
u64 pow_mod(u64 q, u64 base, u64 e);
u64 inverse_mod_prime(u64 elem, u64 prime);        // mod_arith.cpp:136-149
u64 minus_q_inv_mod_2to64(u64 q);
u64 two_to_64_mod(u64 q);
unsigned bit_rev(unsigned x, int bits);
int log_modulus(u64 q);                             // (u64)(log2(q) + 0.5)

ModConsts make_consts(u64 q);

// Returns "" on success, otherwise the message hehub throws for the same input
// ("2N doesn't divide (modulus - 1)" / "NTT not supporting primes with bit size > 59 currently.").
std::string check_ntt_modulus(u64 q, size_t logn);
u64 unity_root_2n(u64 q, size_t logn);              // ntt.cpp:26-39

// Reference-order tables.
//   fwd_ref[i]            = psi^bitrev(i, logn)                       i in [0, N)
//   inv_ref[2^l - 1 + i]  = psi^-(bitrev(i, l) * 2^(logn - l))         l in [0, logn)
//   inv_ref[N + i]        = strict(psi^-i * N^-1)                      i in [0, N)
void build_fwd_ref(u64 q, size_t logn, std::vector<Pair> &out);   // N entries
void build_inv_ref(u64 q, size_t logn, std::vector<Pair> &out);   // 2N entries

// Kernel-order tables for the fast (register + LDS exchange) transforms,
// logn in [11, 15]; layouts documented in hp_ntt_fast.hip.
void build_fwd_fast(const std::vector<Pair> &fwd_ref, size_t logn, std::vector<Pair> &out);
void build_inv_fast(const std::vector<Pair> &inv_ref, size_t logn, std::vector<Pair> &out);

// Parity level A (hp_ntt_a.hip): any of the tables above with every pair (w, w') replaced by the bit patterns of the IEEE doubles
// (w, RN(w / q)); same layout.  q < 2^50 (every w < q is a double exactly).
void pairs_to_f64(const std::vector<Pair> &in, u64 q, std::vector<Pair> &out);
u64 f64_bits(double v);

} // namespace hp
