// hp_device.h -- device-side integer helpers shared by every HIP kernel of the engine.
//
// All arithmetic is wrapping u64 with u128 intermediates, restating the
// per-element formulas of hehub's mod_arith.{h,cpp} / ntt.cpp so that raw lazy
// words are identical to the reference.  gfx950 has no
// 64x64->128 multiply: a wide product is built from v_mad_u64_u32 /
// v_mul_hi_u32 (measured half rate, ~4 cycles per wave64 instruction, see
// tools/ubench/ubench_valu.hip), so the helpers below are written to minimise the
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

// Parity level A (hp_ntt_a.hip): the same per-limb record for the FP64 residue transforms.  Tables have the layouts of
// HpLimb's, every pair (w, w') replaced by the IEEE doubles (w, RN(w / q)) (bit patterns in the u64x2).  q < 2^50.
struct HpLimbA {
    double q;
    double qinv;    // RN(1 / q)
    u64 qi;         // q as an integer
    u32 wide;       // q >= 2^44: coefficients are brought back to |x| <= q/2 between the passes (see hp_ntt_a.hip)
    u32 hi_bound;   // high half of 2 q - 1: a word of a caller's row whose high half is larger is not a lazy word of this limb
    const u64x2 *fwd_ref, *inv_ref, *fwd_k, *inv_k;
    u32 *range_flag;   // the family's sticky "a level-A kernel was handed a word outside its range" flag (hp_ntt_a.hip: RangeAcc)
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
// compiler's 23; 16 in hp_butterfly2_nq, where the low chain also absorbs the addition of the other input).  nq = 2^64 - q is wave-uniform (SGPRs), so x*w - qhat*q becomes the wrapping sum
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

// mod_arith.cpp:9-17 (x - q * floor(x c / 2^64), c = floor((2^64 - 1) / q)) with the same hand-scheduled product chains: the exact
// high product, then x + qhat * (2^64 - q) with the low chain starting from x itself.  10 VALU instructions instead of the
// compiler's ~21 for hp_barrett_lazy; identical value for every u64 x.
HP_DEV u64 hp_barrett_lazy_nq(u64 x, u64 c, u32 n0, u32 n1) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), p0 = (u32)c, p1 = (u32)(c >> 32);
    const u64 A = (u64)x1 * p0 + (u64)__umulhi(x0, p0);
    u64 B, sd;
    u32 cy;
    asm("v_mad_u64_u32 %0, vcc, %3, %4, %5\n\t"
        "s_nop 1\n\t"                                // two wait states between the mad that writes vcc and its reader (as in hp_harvey_lazy_nq)
        "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
        : "=&v"(B), "=&v"(cy), "=&s"(sd)
        : "v"(x0), "v"(p1), "v"(A)
        : "vcc");
    const u64 Q = (u64)x1 * p1 + (((u64)cy << 32) | (B >> 32));
    const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
    u64 G = x, E;
    asm("v_mad_u64_u32 %0, %2, %3, %5, %0\n\t"
        "v_mad_u64_u32 %1, %2, %3, %6, 0\n\t"
        "v_mad_u64_u32 %1, %2, %4, %5, %1"
        : "+v"(G), "=&v"(E), "=&s"(sd)
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

// Two independent butterflies with their instruction streams interleaved by hand.  A single wave
// can start a dependent v_mad_u64_u32 only every ~9 cycles and the SIMD holds just four waves of
// this kernel, so two independent chains per wave are what keeps the half-rate multiplier busy:
// 61 instead of 73 cycles per wave-butterfly per SIMD (tools/ubench/ubench_bfly.hip).
// (wa, wha) and (wb, whb) may be the same twiddle.
HP_DEV void hp_butterfly2_nq(u64 &lo_a, u64 &hi_a, u64 &lo_b, u64 &hi_b, u64 wa, u64 wha, u64 wb, u64 whb, u64 two_q,
                             u32 n0, u32 n1) {
    const u32 ap0 = (u32)wha, ap1 = (u32)(wha >> 32), aw0 = (u32)wa, aw1 = (u32)(wa >> 32);
    const u32 bp0 = (u32)whb, bp1 = (u32)(whb >> 32), bw0 = (u32)wb, bw1 = (u32)(wb >> 32);
    const u32 ax0 = (u32)hi_a, ax1 = (u32)(hi_a >> 32), bx0 = (u32)hi_b, bx1 = (u32)(hi_b >> 32);
    const u64 aA = (u64)ax1 * ap0 + (u64)__umulhi(ax0, ap0);
    const u64 bA = (u64)bx1 * bp0 + (u64)__umulhi(bx0, bp0);
    u64 aB, aG, aE, bB, bG, bE, sd;
    u32 ac, bc;
    // The low-product chains G start from the other butterfly input instead of 0 (wrapping u64, like the reference's
    // x[l] + t): lo + t costs nothing, and hi = lo + 2q - t becomes (2 lo + 2q) - (lo + t).
    asm("v_mad_u64_u32 %0, vcc, %9, %10, %11\n\t"     // aB = ax0*ap1 + aA, carry -> vcc
        "v_mad_u64_u32 %1, %8, %9, %12, %21\n\t"      // aG = ax0*aw0 + lo_a
        "v_mad_u64_u32 %2, %8, %9, %13, 0\n\t"        // aE = ax0*aw1
        "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"    // ac
        "v_mad_u64_u32 %4, vcc, %15, %18, %16\n\t"    // bB = bx0*bp1 + bA, carry -> vcc
        "v_mad_u64_u32 %2, %8, %14, %12, %2\n\t"      // aE += ax1*aw0
        "v_mad_u64_u32 %5, %8, %15, %19, %22\n\t"     // bG = bx0*bw0 + lo_b
        "v_addc_co_u32_e64 %7, vcc, 0, 0, vcc\n\t"    // bc
        "v_mad_u64_u32 %6, %8, %15, %20, 0\n\t"       // bE = bx0*bw1
        "v_mad_u64_u32 %6, %8, %17, %19, %6"            // bE += bx1*bw0
        : "=&v"(aB), "=&v"(aG), "=&v"(aE), "=&v"(ac), "=&v"(bB), "=&v"(bG), "=&v"(bE), "=&v"(bc), "=&s"(sd)
        : "v"(ax0), "v"(ap1), "v"(aA), "v"(aw0), "v"(aw1), "v"(ax1), "v"(bx0), "v"(bA), "v"(bx1), "v"(bp1), "v"(bw0),
          "v"(bw1), "v"(lo_a), "v"(lo_b)
        : "vcc");
    const u64 aU = ((u64)ac << 32) | (aB >> 32), bU = ((u64)bc << 32) | (bB >> 32);
    const u64 aQ = (u64)ax1 * ap1 + aU, bQ = (u64)bx1 * bp1 + bU;
    const u32 aq0 = (u32)aQ, aq1 = (u32)(aQ >> 32), bq0 = (u32)bQ, bq1 = (u32)(bQ >> 32);
    asm("v_mad_u64_u32 %0, %4, %5, %10, %0\n\t"
        "v_mad_u64_u32 %2, %4, %7, %10, %2\n\t"
        "v_mad_u64_u32 %1, %4, %5, %9, %1\n\t"
        "v_mad_u64_u32 %3, %4, %7, %9, %3\n\t"
        "v_mad_u64_u32 %0, %4, %6, %9, %0\n\t"
        "v_mad_u64_u32 %2, %4, %8, %9, %2"
        : "+v"(aE), "+v"(aG), "+v"(bE), "+v"(bG), "=&s"(sd)
        : "v"(aq0), "v"(aq1), "v"(bq0), "v"(bq1), "s"(n0), "s"(n1));
    u32 ath, bth;
    asm("v_add_u32 %0, %1, %2" : "=v"(ath) : "v"((u32)(aG >> 32)), "v"((u32)aE));
    asm("v_add_u32 %0, %1, %2" : "=v"(bth) : "v"((u32)(bG >> 32)), "v"((u32)bE));
    const u64 as = ((u64)ath << 32) | (u32)aG, bs = ((u64)bth << 32) | (u32)bG;   // lo + t
    hi_a = ((lo_a << 1) + two_q) - as;
    lo_a = as;
    hi_b = ((lo_b << 1) + two_q) - bs;
    lo_b = bs;
}

// The same with the twiddle words in SGPRs (first pass of a transform: wave-uniform twiddles from scalar loads): every multiply
// of the block names at most one SGPR, so the words need not be copied into VGPRs (124 v_mov per thread in the first pass).
HP_DEV void hp_butterfly2_nq_sw(u64 &lo_a, u64 &hi_a, u64 &lo_b, u64 &hi_b, u64 wa, u64 wha, u64 wb, u64 whb, u64 two_q,
                             u32 n0, u32 n1) {
    const u32 ap0 = (u32)wha, ap1 = (u32)(wha >> 32), aw0 = (u32)wa, aw1 = (u32)(wa >> 32);
    const u32 bp0 = (u32)whb, bp1 = (u32)(whb >> 32), bw0 = (u32)wb, bw1 = (u32)(wb >> 32);
    const u32 ax0 = (u32)hi_a, ax1 = (u32)(hi_a >> 32), bx0 = (u32)hi_b, bx1 = (u32)(hi_b >> 32);
    const u64 aA = (u64)ax1 * ap0 + (u64)__umulhi(ax0, ap0);
    const u64 bA = (u64)bx1 * bp0 + (u64)__umulhi(bx0, bp0);
    u64 aB, aG, aE, bB, bG, bE, sd;
    u32 ac, bc;
    // The low-product chains G start from the other butterfly input instead of 0 (wrapping u64, like the reference's
    // x[l] + t): lo + t costs nothing, and hi = lo + 2q - t becomes (2 lo + 2q) - (lo + t).
    asm("v_mad_u64_u32 %0, vcc, %9, %10, %11\n\t"     // aB = ax0*ap1 + aA, carry -> vcc
        "v_mad_u64_u32 %1, %8, %9, %12, %21\n\t"      // aG = ax0*aw0 + lo_a
        "v_mad_u64_u32 %2, %8, %9, %13, 0\n\t"        // aE = ax0*aw1
        "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"    // ac
        "v_mad_u64_u32 %4, vcc, %15, %18, %16\n\t"    // bB = bx0*bp1 + bA, carry -> vcc
        "v_mad_u64_u32 %2, %8, %14, %12, %2\n\t"      // aE += ax1*aw0
        "v_mad_u64_u32 %5, %8, %15, %19, %22\n\t"     // bG = bx0*bw0 + lo_b
        "v_addc_co_u32_e64 %7, vcc, 0, 0, vcc\n\t"    // bc
        "v_mad_u64_u32 %6, %8, %15, %20, 0\n\t"       // bE = bx0*bw1
        "v_mad_u64_u32 %6, %8, %17, %19, %6"            // bE += bx1*bw0
        : "=&v"(aB), "=&v"(aG), "=&v"(aE), "=&v"(ac), "=&v"(bB), "=&v"(bG), "=&v"(bE), "=&v"(bc), "=&s"(sd)
        : "v"(ax0), "s"(ap1), "v"(aA), "s"(aw0), "s"(aw1), "v"(ax1), "v"(bx0), "v"(bA), "v"(bx1), "s"(bp1), "s"(bw0),
          "s"(bw1), "v"(lo_a), "v"(lo_b)
        : "vcc");
    const u64 aU = ((u64)ac << 32) | (aB >> 32), bU = ((u64)bc << 32) | (bB >> 32);
    const u64 aQ = (u64)ax1 * ap1 + aU, bQ = (u64)bx1 * bp1 + bU;
    const u32 aq0 = (u32)aQ, aq1 = (u32)(aQ >> 32), bq0 = (u32)bQ, bq1 = (u32)(bQ >> 32);
    asm("v_mad_u64_u32 %0, %4, %5, %10, %0\n\t"
        "v_mad_u64_u32 %2, %4, %7, %10, %2\n\t"
        "v_mad_u64_u32 %1, %4, %5, %9, %1\n\t"
        "v_mad_u64_u32 %3, %4, %7, %9, %3\n\t"
        "v_mad_u64_u32 %0, %4, %6, %9, %0\n\t"
        "v_mad_u64_u32 %2, %4, %8, %9, %2"
        : "+v"(aE), "+v"(aG), "+v"(bE), "+v"(bG), "=&s"(sd)
        : "v"(aq0), "v"(aq1), "v"(bq0), "v"(bq1), "s"(n0), "s"(n1));
    u32 ath, bth;
    asm("v_add_u32 %0, %1, %2" : "=v"(ath) : "v"((u32)(aG >> 32)), "v"((u32)aE));
    asm("v_add_u32 %0, %1, %2" : "=v"(bth) : "v"((u32)(bG >> 32)), "v"((u32)bE));
    const u64 as = ((u64)ath << 32) | (u32)aG, bs = ((u64)bth << 32) | (u32)bG;   // lo + t
    hi_a = ((lo_a << 1) + two_q) - as;
    lo_a = as;
    hi_b = ((lo_b << 1) + two_q) - bs;
    lo_b = bs;
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

// ---- 128-bit multiply-accumulate in carry-save columns ------------------------------------------
// sum_j x_j * y_j mod 2^128 (the u128 accumulators of rgsw.cpp:126-149) as three 64-bit columns
//     p00 = sum x0*y0,   px = sum (x0*y1 + x1*y0)  (weight 2^32),   p11 = sum x1*y1  (weight 2^64)
// plus the counts c00 / cx of the carries out of p00 / px; what p11 loses has weight 2^128.  One term costs the four
// v_mad_u64_u32 and three v_addc_co (7 VALU instructions; the compiler's u128 += u64*u64 is ~20).  Exact for any u64 words.
struct HpAcc {
    u64 p00, px, p11;
    u32 c00, cx;
};
HP_DEV void hp_acc_zero(HpAcc &a) { a.p00 = a.px = a.p11 = 0; a.c00 = a.cx = 0; }
// two independent accumulators in one block: the carry of a mad (an SGPR pair) is consumed four instructions later
HP_DEV void hp_mac2(HpAcc &a, u64 ax, u64 ay, HpAcc &b, u64 bx, u64 by) {
    const u32 ax0 = (u32)ax, ax1 = (u32)(ax >> 32), ay0 = (u32)ay, ay1 = (u32)(ay >> 32);
    const u32 bx0 = (u32)bx, bx1 = (u32)(bx >> 32), by0 = (u32)by, by1 = (u32)(by >> 32);
    u64 s0, s1, s2, s3, sd;
    asm("v_mad_u64_u32 %0, %10, %15, %17, %0\n\t"     // a.p00 += ax0*ay0      carry -> s0
        "v_mad_u64_u32 %1, %11, %15, %18, %1\n\t"     // a.px  += ax0*ay1      carry -> s1
        "v_mad_u64_u32 %5, %12, %19, %21, %5\n\t"     // b.p00 += bx0*by0      carry -> s2
        "v_mad_u64_u32 %6, %13, %19, %22, %6\n\t"     // b.px  += bx0*by1      carry -> s3
        "v_addc_co_u32_e64 %3, %14, %3, 0, %10\n\t"   // a.c00 += s0
        "v_addc_co_u32_e64 %4, %14, %4, 0, %11\n\t"   // a.cx  += s1
        "v_addc_co_u32_e64 %8, %14, %8, 0, %12\n\t"   // b.c00 += s2
        "v_addc_co_u32_e64 %9, %14, %9, 0, %13\n\t"   // b.cx  += s3
        "v_mad_u64_u32 %1, %11, %16, %17, %1\n\t"     // a.px  += ax1*ay0      carry -> s1
        "v_mad_u64_u32 %6, %13, %20, %21, %6\n\t"     // b.px  += bx1*by0      carry -> s3
        "v_mad_u64_u32 %2, %14, %16, %18, %2\n\t"     // a.p11 += ax1*ay1
        "v_mad_u64_u32 %7, %14, %20, %22, %7\n\t"     // b.p11 += bx1*by1
        "v_addc_co_u32_e64 %4, %14, %4, 0, %11\n\t"   // a.cx  += s1
        "v_addc_co_u32_e64 %9, %14, %9, 0, %13"          // b.cx  += s3
        : "+v"(a.p00), "+v"(a.px), "+v"(a.p11), "+v"(a.c00), "+v"(a.cx), "+v"(b.p00), "+v"(b.px), "+v"(b.p11), "+v"(b.c00),
          "+v"(b.cx), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(sd)
        : "v"(ax0), "v"(ax1), "v"(ay0), "v"(ay1), "v"(bx0), "v"(bx1), "v"(by0), "v"(by1));
}
// the accumulated value: lo = p00 + (px << 32), hi = c00 + (px >> 32) + (cx << 32) + p11 + carry(lo)
HP_DEV void hp_acc_value(const HpAcc &a, u64 &lo, u64 &hi) {
    const u64 sh = a.px << 32;
    lo = a.p00 + sh;
    hi = (u64)a.c00 + (a.px >> 32) + ((u64)a.cx << 32) + a.p11 + (lo < sh ? 1ull : 0ull);
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

// HP_PACK48: row format of the digit workspace for output moduli whose words are provably below 2^48 (internal to the engine:
// written by the digit-spread transform, read by the key-switch inner product; 6 instead of 8 bytes per word cross HBM in both
// directions).  A row keeps its 8N-byte slot: [u32 lo[N]] [u16 hi[N]] [2N bytes unused]; word i = lo[i] | (u64)hi[i] << 32.
// (One 12-byte record per pair of words, moved with dwordx3 accesses, measured no faster than plain rows: the accesses straddle
// cache lines.  Two planes: -13..-16 % on the inner product at every tiled ring degree once that kernel addressed its rows through
// buffer descriptors, -1.5 % on the spread launch at N = 32768.)
// HP_PACK40 (parity level A only, where a digit row may hold any representative): for a modulus with q + 2 <= 2^40 the row keeps
// r + (q - 1)/2 + 1 with r the centred residue (|r| <= (q - 1)/2 + 1 after x - rint(x / q) q), a value in [0, 2^40):
// [u32 lo[N]] [u8 hi[N]] [3N bytes unused], 5 of 8 bytes cross HBM.  The inner product adds (q - 1)/2 times the sum of the key words
// it multiplied the rows by: sum (w + (q - 1)/2) key = sum (r + q) key.
// XCD-aware work-item remap: consecutive blockIdx values land on different XCDs
// (block b -> XCD b % 8, observed placement; used for L2 locality only).  Work
// items are numbered so that neighbours share a modulus (twiddle table); this
// map hands XCD x the contiguous slice [x*W/8, (x+1)*W/8).
HP_DEV u32 hp_xcd_remap(u32 b, u32 W) {
    const u32 per = W >> 3;
    if (b >= (per << 3)) return b; // tail that does not divide evenly
    return (b & 7u) * per + (b >> 3);
}
