// hp_tables.cpp -- see hp_tables.h
#include "hp_tables.h"
#include <cstring>

#include <cmath>

namespace hp {

u64 harvey_quotient(u64 b, u64 q) { return (u64)(((u128)b << 64) / q); }

u64 mul_mod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

u64 pow_mod(u64 q, u64 base, u64 e) {
    u64 result = 1 % q;
    u64 sq = base % q;
    while (e) {
        if (e & 1) result = mul_mod(result, sq, q);
        sq = mul_mod(sq, sq, q);
        e >>= 1;
    }
    return result;
}

// Bezout coefficient of elem in prime*x + elem*y = 1, lifted to [0, prime)
// (what hehub's xgcd128-based inverse_mod_prime returns, mod_arith.cpp:136-149).
u64 inverse_mod_prime(u64 elem, u64 prime) {
    __int128 r0 = prime, r1 = elem, y0 = 0, y1 = 1;
    while (r1 != 0) {
        __int128 quo = r0 / r1;
        __int128 r2 = r0 - quo * r1, y2 = y0 - quo * y1;
        r0 = r1; r1 = r2; y0 = y1; y1 = y2;
    }
    if (y0 < 0) y0 += prime;
    return (u64)y0;
}

u64 minus_q_inv_mod_2to64(u64 q) {
    u64 inv = q;
    for (int i = 0; i < 6; i++) inv *= 2 - q * inv;
    return (u64)0 - inv;
}

u64 two_to_64_mod(u64 q) { return (~(u64)0) % q + 1; }

unsigned bit_rev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

int log_modulus(u64 q) { return (int)(u64)(std::log2((double)q) + 0.5); }

ModConsts make_consts(u64 q) {
    ModConsts c;
    c.q = q;
    c.two_q = 2 * q;
    c.neg_q = (u64)0 - q;
    c.mqinv = (q & 1) ? minus_q_inv_mod_2to64(q) : 0;
    c.r64 = two_to_64_mod(q);
    c.r64h = harvey_quotient(c.r64 % q, q); // r64 == q only if q | 2^64
    c.barrett_c = (~(u64)0) / q;
    c.k = (uint32_t)log_modulus(q);
    c.fix = (c.k < 64 && q >= ((u64)1 << c.k)) ? 1u : 0u;
    return c;
}

std::string check_ntt_modulus(u64 q, size_t logn) {
    if (log_modulus(q) > 59) return "NTT not supporting primes with bit size > 59 currently.";
    if ((q - 1) % ((u64)2 << logn) != 0) return "2N doesn't divide (modulus - 1)";
    return "";
}

u64 unity_root_2n(u64 q, size_t logn) {
    u64 g = 2;
    while (pow_mod(q, g, (q - 1) / 2) != q - 1) g++;
    return pow_mod(q, g, (q - 1) / ((u64)2 << logn));
}

static void powers(u64 q, u64 base, size_t count, std::vector<u64> &pw) {
    pw.resize(count);
    u64 cur = 1;
    for (size_t e = 0; e < count; e++) {
        pw[e] = cur;
        cur = mul_mod(cur, base, q);
    }
}

void build_fwd_ref(u64 q, size_t logn, std::vector<Pair> &out) {
    const size_t n = (size_t)1 << logn;
    std::vector<u64> pw;
    powers(q, unity_root_2n(q, logn), n, pw);
    out.resize(n);
    for (size_t i = 0; i < n; i++) {
        u64 w = pw[bit_rev((unsigned)i, (int)logn)];
        out[i] = Pair{w, harvey_quotient(w, q)};
    }
}

void build_inv_ref(u64 q, size_t logn, std::vector<Pair> &out) {
    const size_t n = (size_t)1 << logn;
    const u64 psi = unity_root_2n(q, logn);
    const u64 psi_inv = pow_mod(q, psi, 2 * n - 1);
    std::vector<u64> pw;
    powers(q, psi_inv, n, pw);
    out.assign(2 * n, Pair{0, 0});
    for (size_t l = 0; l < logn; l++) {
        const size_t start = ((size_t)1 << l) - 1;
        for (size_t i = 0; i < ((size_t)1 << l); i++) {
            u64 w = pw[(size_t)bit_rev((unsigned)i, (int)l) << (logn - l)];
            out[start + i] = Pair{w, harvey_quotient(w, q)};
        }
    }
    const u64 n_inv = q - ((q - 1) >> logn);
    const u64 n_inv_h = harvey_quotient(n_inv, q);
    for (size_t i = 0; i < n; i++) {
        u64 qhat = (u64)(((u128)pw[i] * n_inv_h) >> 64);
        u64 w = (u64)((u128)pw[i] * n_inv - (u128)qhat * q);
        w -= (w >= q) ? q : 0;
        out[n + i] = Pair{w, harvey_quotient(w, q)};
    }
}

// ---- kernel-order tables (layout contract with hp_ntt_fast.hip) -------------
//
// Every pass of the fast kernels runs up to five radix-2 stages on the 5-bit
// register index r of a thread.  A stage works on register bit b and pairs
// r (bit b clear) with r | (1 << b).
//
// forward: stages go b = 4,3,2,1,0; the twiddle of a pair depends on
//          sub = r >> (b + 1) (the bits above b); slot = (2^(4-b) - 1) + sub.
// inverse: stages go b = 0,1,2,3,4; the twiddle depends on
//          low = r & (2^b - 1) (the bits below b); slot = (2^b - 1) + low.
// A pass table is laid out [slot][class] so that the lanes of a wavefront read
// consecutive 16-byte pairs.

void build_fwd_fast(const std::vector<Pair> &fwd_ref, size_t logn, std::vector<Pair> &out) {
    const size_t a = logn - 10;            // stages done by pass A
    const size_t nblk = (size_t)1 << a;    // 1024-coefficient blocks
    const size_t T = (size_t)1 << (logn - 5);
    out.assign(31 * nblk + 31 * T, Pair{0, 0});
    Pair *B = out.data();
    Pair *C = out.data() + 31 * nblk;
    for (size_t sp = 1; sp <= 5; sp++) {               // stage within the pass
        for (size_t sub = 0; sub < ((size_t)1 << (sp - 1)); sub++) {
            const size_t slot = ((size_t)1 << (sp - 1)) - 1 + sub;
            for (size_t blk = 0; blk < nblk; blk++)    // pass B: global stage a + sp
                B[slot * nblk + blk] = fwd_ref[((size_t)1 << (a + sp - 1)) + (blk << (sp - 1)) + sub];
            for (size_t t = 0; t < T; t++)             // pass C: global stage a + 5 + sp
                C[slot * T + t] = fwd_ref[((size_t)1 << (a + 5 + sp - 1)) + (t << (sp - 1)) + sub];
        }
    }
}

void build_inv_fast(const std::vector<Pair> &inv_ref, size_t logn, std::vector<Pair> &out) {
    const size_t a = logn - 10;
    const size_t T = (size_t)1 << (logn - 5);
    out.assign(31 + 31 * 32 + 31 * T, Pair{0, 0});
    Pair *IA = out.data();
    Pair *IB = IA + 31;
    Pair *IC = IB + 31 * 32;
    for (size_t b = 0; b < 5; b++) {
        for (size_t low = 0; low < ((size_t)1 << b); low++) {
            const size_t slot = ((size_t)1 << b) - 1 + low;
            {   // pass A': level l = b, c = low
                const size_t l = b;
                IA[slot] = inv_ref[((size_t)1 << l) - 1 + bit_rev((unsigned)low, (int)l)];
            }
            for (size_t j = 0; j < 32; j++) {   // pass B': level l = 5 + b, c = (low << 5) | j
                const size_t l = 5 + b;
                const size_t c = (low << 5) | j;
                IB[slot * 32 + j] = inv_ref[((size_t)1 << l) - 1 + bit_rev((unsigned)c, (int)l)];
            }
            if (b >= 5 - a) {                   // pass C': level l = 10 + (b - (5 - a))
                const size_t l = 10 + (b - (5 - a));
                const size_t kk_low = low >> (5 - a);
                const size_t pp = low & (((size_t)1 << (5 - a)) - 1);
                for (size_t t = 0; t < T; t++) {
                    const size_t c = (kk_low << 10) | (t << (5 - a)) | pp;
                    IC[slot * T + t] = inv_ref[((size_t)1 << l) - 1 + bit_rev((unsigned)c, (int)l)];
                }
            }
        }
    }
}


u64 f64_bits(double v) {
    u64 b;
    static_assert(sizeof(b) == sizeof(v), "IEEE double");
    memcpy(&b, &v, sizeof(b));
    return b;
}

void pairs_to_f64(const std::vector<Pair> &in, u64 q, std::vector<Pair> &out) {
    out.resize(in.size());
    const double qd = (double)q;
    for (size_t i = 0; i < in.size(); i++) {
        const double w = (double)in[i].w;   // exact: w < q < 2^50
        out[i] = Pair{f64_bits(w), f64_bits(w / qd)};
    }
}

} // namespace hp
