// hp_device.h -- device-side integer helpers shared by every HIP kernel of the engine.
//
// All arithmetic is wrapping u64 with u128 intermediates, restating the
// per-element formulas of hehub's mod_arith.{h,cpp} / ntt.cpp so that raw lazy
// words are identical to the reference.  gfx950 has no
// 64x64->128 multiply: a wide product is built from v_mad_u64_u32 /
// v_mul_hi_u32 (measured half rate, ~4 cycles per wave64 instruction, see
// tools/ubench_valu.hip), so the helpers below are written to minimise the
// number of 32-bit multiplies and to keep 64-bit additions on v_lshl_add_u64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

struct u64x2 {
    u64 x, y;
};

// Per-(modulus, logN) constants living in device memory; one entry per RNS limb of a "plan".
struct HpLimb {
    u64 q;
    u64 two_q;
    u64 neg_q;      // 2^64 - q
    u64 mqinv;      // -q^{-1} mod 2^64          (mod_arith.cpp:49-52)
    u64 r64;        // 2^64 mod q                 (mod_arith.cpp:54-57)
    u64 r64h;       // floor(r64 * 2^64 / q)      (mod_arith.cpp:59-62)
    u64 barrett_c;  // floor((2^64-1)/q)          (mod_arith.cpp:10)
    u32 k;          // (u64)(log2(q)+0.5)         (ntt.cpp:171)
    u32 fix;        // q >= 2^k                   (ntt.cpp:172)
    const u64x2 *fwd_ref;   // forward (w, w') pairs, reference order seq[idx], N entries (ntt.cpp:49-58)
    const u64x2 *inv_ref;   // inverse pairs in reference order, 2N entries (ntt.cpp:59-85)
    const u64x2 *fwd_k;     // forward pairs in fast-kernel order (see hp_ntt_fast.hip)
    const u64x2 *inv_k;     // inverse pairs in fast-kernel order
};

#define HP_DEV __device__ __forceinline__

HP_DEV u64 hp_mulhi(u64 a, u64 b) { return __umul64hi(a, b); }

// mod_arith.h:74-78
HP_DEV u64 hp_harvey_lazy(u64 x, u64 w, u64 wh, u64 q) {
    u64 qhat = hp_mulhi(x, wh);
    return x * w - qhat * q;
}

// the radix-2 lazy butterfly of ntt.cpp:160-166
HP_DEV void hp_butterfly(u64 &lo, u64 &hi, u64 w, u64 wh, u64 q, u64 two_q) {
    u64 t = hp_harvey_lazy(hi, w, wh, q);
    hi = lo + two_q - t;
    lo = lo + t;
}

// The same product, hand-scheduled for gfx950 (17 VALU instructions per butterfly instead of the
// compiler's 23).  nq = 2^64 - q is wave-uniform (SGPRs), so x*w - qhat*q becomes the wrapping sum
// x*w + qhat*nq and both low products chain through v_mad_u64_u32 accumulators:
//   qhat = floor(x*wh / 2^64), exactly:   A = x1*p0 + hi32(x0*p0)
//                                          B = x0*p1 + A           (carry c from the mad's carry-out)
//                                          qhat = x1*p1 + {B.hi, c}
//   t.lo64 = (x0*w0 + q0*n0) + 2^32 * lo32(x0*w1 + x1*w0 + q0*n1 + q1*n0)
// The asm statements only ever name whole registers; halves are split in C (free sub-register
// references).  Hazards inside the strings: the v_addc_co reading the carry sits two instructions
// after the v_mad_u64_u32 that writes vcc (the spacing the compiler itself uses on gfx950).
HP_DEV u64 hp_harvey_lazy_nq(u64 x, u64 w, u64 wh, u32 n0, u32 n1) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), p0 = (u32)wh, p1 = (u32)(wh >> 32);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
    const u64 A = (u64)x1 * p0 + (u64)__umulhi(x0, p0);
    u64 B, G, E, sd;
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %5, %6, %7\n\t"
        "v_mad_u64_u32 %1, %4, %5, %8, 0\n\t"
        "v_mad_u64_u32 %2, %4, %5, %9, 0\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
        "v_mad_u64_u32 %2, %4, %10, %8, %2"
        : "=&v"(B), "=&v"(G), "=&v"(E), "=&v"(c), "=&s"(sd)
        : "v"(x0), "v"(p1), "v"(A), "v"(w0), "v"(w1), "v"(x1)
        : "vcc");
    const u64 U = ((u64)c << 32) | (B >> 32);
    const u64 Q = (u64)x1 * p1 + U;
    const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
    asm("v_mad_u64_u32 %0, %2, %3, %6, %0\n\t"
        "v_mad_u64_u32 %1, %2, %3, %5, %1\n\t"
        "v_mad_u64_u32 %0, %2, %4, %5, %0"
        : "+v"(E), "+v"(G), "=&s"(sd)
        : "v"(q0), "v"(q1), "s"(n0), "s"(n1));
    u32 th;
    asm("v_add_u32 %0, %1, %2" : "=v"(th) : "v"((u32)(G >> 32)), "v"((u32)E));
    return ((u64)th << 32) | (u32)G;
}

HP_DEV void hp_butterfly_nq(u64 &lo, u64 &hi, u64 w, u64 wh, u64 two_q, u32 n0, u32 n1) {
    const u64 t = hp_harvey_lazy_nq(hi, w, wh, n0, n1);
    hi = lo + two_q - t;
    lo = lo + t;
}

// ntt.cpp:171-175
HP_DEV u64 hp_shift_fold(u64 x, u64 q, u32 k, u32 fix) { return x - ((x >> k) - (u64)fix) * q; }

// mod_arith.h:58-63
HP_DEV u64 hp_strict(u64 x, u64 q) { return x - ((x >= q) ? q : 0); }

// mod_arith.cpp:9-17
HP_DEV u64 hp_barrett_lazy(u64 x, u64 q, u64 c) { return x - q * hp_mulhi(x, c); }

// 64x64 -> 128
HP_DEV void hp_mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    lo = a * b;
    hi = hp_mulhi(a, b);
}

// mod_arith.cpp:113-134: (acc + ((acc.lo * m) mod 2^64) * q) >> 64
HP_DEV u64 hp_montgomery128_lazy(u64 lo, u64 hi, u64 q, u64 mqinv) {
    u64 u = lo * mqinv;
    // low word of acc + u*q is zero by construction; its carry is (lo != 0)
    u64 uq_hi = hp_mulhi(u, q);
    return hi + uq_hi + (lo != 0 ? 1ull : 0ull);
}

// mod_arith.cpp:64-92
HP_DEV u64 hp_mul_hybrid_lazy(u64 a, u64 b, const HpLimb &m) {
    u64 lo, hi;
    hp_mul128(a, b, lo, hi);
    u64 t = hp_montgomery128_lazy(lo, hi, m.q, m.mqinv);
    return hp_harvey_lazy(t, m.r64, m.r64h, m.q);
}

// rns.cpp:78-84
HP_DEV u64 hp_add_lazy(u64 a, u64 b, u64 two_q) {
    u64 v = a + b;
    return v - ((v >= two_q) ? two_q : 0);
}

// rns.cpp:109-115
HP_DEV u64 hp_sub_lazy(u64 a, u64 b, u64 two_q) {
    u64 v = a + (two_q - b);
    return v - ((v >= two_q) ? two_q : 0);
}

// XCD-aware work-item remap: consecutive blockIdx values land on different XCDs
// (block b -> XCD b % 8, observed placement; used for L2 locality only).  Work
// items are numbered so that neighbours share a modulus (twiddle table); this
// map hands XCD x the contiguous slice [x*W/8, (x+1)*W/8).
HP_DEV u32 hp_xcd_remap(u32 b, u32 W) {
    const u32 per = W >> 3;
    if (b >= (per << 3)) return b; // tail that does not divide evenly
    return (b & 7u) * per + (b >> 3);
}
