// hp_ntt_a.hip -- parity level A (SURVEY.md section 8 "Parity levels"; opt-in through hp_ctx_set_parity_level): the tiled
// negacyclic NTT / INTT of hp_ntt_fast.hip with the butterfly's modular product computed in error-free FP64 arithmetic instead of
// the exact Harvey quotient.  Every output word is the CANONICAL residue in [0, q): congruent to the reference's lazy word
// (ntt.cpp:155-175, :178-223, rescaling.cpp:46-75, mod_switch.cpp:45-77) and equal to reduce_strict of it -- not the same
// representative, which is why this is level A and off by default.
//
// Why: the level-B butterfly is 16 integer VALU instructions (10 of them 32-bit multiplies) = 62 cycles per wave-butterfly and the
// transforms are bound by exactly that.  For q < 2^50 the residue of x w needs 8 FP64 instructions
// (tools/ubench/ubench_bfly_f64.hip: 31.5 cycles, exact on 2 x 10^9 random and edge cases):
//     h = RN(x w)            l = fma(x, w, -h)             x w = h + l exactly (error-free product)
//     k = rint(x u)          u = RN(w / q), precomputed    |k - x w / q| <= 1/2 + |x| 2^-52
//     t = fma(-k, q, h) + l                                 = x w - k q exactly, |t| <= q (1/2 + |x| 2^-52)
//     hi' = lo - t           lo' = lo + t
// All values are integers held exactly in doubles (signed, |x| < 2^53).  Growth: a stage adds at most |t| to a coefficient.
//   narrow limbs (q < 2^44): inputs below 2^50 (strict residues of any modulus of the chain, or centred last-limb coefficients)
//     stay below 2^51 through 15 stages -- no reduction until the end;
//   wide limbs (2^44 <= q < 2^50): x <- x - rint(x / q) q (3 instructions, |x| <= q/2 afterwards) after every pass: a pass of five
//     stages takes |x| <= q/2 to < 5.7 q < 2^53 (recurrence B' = B (1 + q 2^-52) + q/2), and 2^50 to < 7.2 * 2^50 in the first pass.
// The 64-bit patterns of the doubles travel through the same registers, LDS exchanges and table layouts as the integer kernels
// (hp_ntt_tile.h); tables hold (w, u) as doubles (hp_tables.cpp: pairs_to_f64).  HBM rows stay u64 words: converted on load
// (words must be below 2^52: every lazy word hehub or this engine produces is below 2 q <= 2^51) and on store.
#include "hp_ntt_tile.h"

#pragma clang fp contract(off)

namespace {

typedef const HpLimbA __attribute__((address_space(4))) * cptr_limba;
#ifdef HP_TRACE
__device__ u64 g_trace[2 * 2048 * 16 * HP_TRACE_SLOTS];
#endif

HP_DEV double D(u64 v) { return __builtin_bit_cast(double, v); }
HP_DEV u64 U(double d) { return __builtin_bit_cast(u64, d); }
constexpr double TWO52 = 4503599627370496.0;
HP_DEV double from_word(u64 w) { return D(w | 0x4330000000000000ull) - TWO52; }        // w < 2^52
HP_DEV u64 to_word(double v) { return U(v + TWO52) & 0x000FFFFFFFFFFFFFull; }           // integral 0 <= v < 2^52

// x w - rint(x w / q) q, exactly (|x| < 2^52, 0 <= w < q < 2^50, u = RN(w / q))
HP_DEV double a_modmul(double x, double w, double u, double q) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double k = __builtin_rint(x * u);
    const double r = __builtin_fma(-k, q, h);
    return r + l;
}
HP_DEV double a_reduce(double x, double qinv, double q) { return __builtin_fma(-__builtin_rint(x * qinv), q, x); }   // |result| <= q/2 (+ 1 ulp of the quotient)
HP_DEV double a_nonneg(double v, double q) { return v < 0.0 ? v + q : v; }                                          // (-q, q) -> [0, q)
HP_DEV u64 a_canon(double x, double qinv, double q) { return to_word(a_nonneg(a_reduce(x, qinv, q), q)); }

HP_DEV void a_bfly(u64 &lo, u64 &hi, double w, double u, double q) {
    const double xl = D(lo), t = a_modmul(D(hi), w, u, q);
    hi = U(xl - t);
    lo = U(xl + t);
}
HP_DEV void a_reduce_all(u64 (&x)[32], double qinv, double q) {
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = U(a_reduce(D(x[r]), qinv, q));
}

// ---- range guard ---------------------------------------------------------------------------------------------------------------
// The conversion of a word to a double (from_word) is exact below 2^52 and the growth bounds above assume lazy words (below 2 q of
// the limb): hehub's own transforms take any u64 (ntt.cpp:155-175) and level B reproduces that, level A cannot.  Every word a
// kernel of this file loads from a row a CALLER may have supplied (the plain forward and inverse transforms, the rows x and addend
// of the drop epilogue) goes through one v_max_u32 on its high half; a thread that saw a high half above that of 2 q - 1 sets the
// family's sticky flag, and the next synchronising call of the C ABI (hp_sync, hp_memcpy_d2h) returns HP_ERANGE (hp_ctx.cpp).
struct RangeAcc {
    u32 m = 0;
    HP_DEV void see(u64 w) { m = max(m, hi32(w)); }
    HP_DEV void report(cptr_limba lp) const {
        if (m > lp->hi_bound) atomicOr(lp->range_flag, 1u);
    }
};

// ---- load-side work of the forward kernels, done per 16-byte load right before the first butterfly that touches it ----------
// F64: the row already holds doubles (the strict coefficient rows of the key switch, written by k_ntt_inv_a with job.dst_f64)
// words (a caller's row): range guard, and the word is brought to |x| <= q/2 at once -- a lazy word of a 50-bit modulus (up to
// 2^51) would otherwise leave the exact range within the first pass (B' = B (1 + q 2^-52) + q/2 from 2^51: 10.2 * 2^50)
// GUARD off: strict coefficient rows of ANOTHER modulus of the chain (the digit spread from words): any residue below 2^50 is in
// range, the caller vouches for "strict" (hp_dev_ks_inner_range_strict) and the bound of the TARGET limb would be the wrong one
template <bool SWAP, bool F64, bool GUARD = true> struct ConvPre {
    static constexpr bool on = SWAP || !F64;
    double q = 0.0, qinv = 0.0;
    mutable RangeAcc acc;
    HP_DEV void operator()(u64 (&x)[32], int r) const {
        if (SWAP) lazy_swap(x, r);
        if (!F64) {
            if (GUARD) {
                acc.see(x[r]);
                acc.see(x[r + 1]);
            }
            // (unconditionally: a uniform branch on `wide` here costs the small ring degrees 6 .. 16 spilled registers, three FP64
            // instructions per word cost a narrow limb 3 % of this kernel -- which no scheme-level pipeline launches)
            x[r] = U(a_reduce(from_word(x[r]), qinv, q));
            x[r + 1] = U(a_reduce(from_word(x[r + 1]), qinv, q));
        }
    }
};
// fused drop-last-prime: c (strict modulo q_last) -> the centred representative c - [c >= q_last / 2] q_last, which is congruent
// modulo q_k to the remainder rescaling.cpp:54-69 builds (Barrett_k(c) + [c >= q_last/2] (q_k - q_last mod q_k)); BGV: times t
template <bool SWAP, bool BGV> struct DropPreA {
    static constexpr bool on = true;
    double q, q_last, half, tw, tu;
    HP_DEV void operator()(u64 (&x)[32], int r) const {
        if (SWAP) lazy_swap(x, r);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double c = from_word(x[r + e]);
            c = (c >= half) ? c - q_last : c;
            if (BGV) c = a_modmul(c, tw, tu, q);
            x[r + e] = U(c);
        }
    }
};

// Two drops in one transform (level A only: residues, not representatives).  Dropping p and then q' from rows x with addend a,
//     y_k = (x_k - NTT_k(cp)) p^-1 + a_k            cp = centred INTT_p(x_p)                  (rescaling.cpp:46-75, first call)
//     z_k = (y_k - NTT_k(cq)) q'^-1                 cq = centred INTT_q'(y_q')                (second call)
// is, by linearity of the transform modulo q_k,
//     z_k = ((A_k x_k + a_k) - NTT_k(m_k cp + m2_k cq)) B_k
// with A_k = m_k = p^-1, m2_k = 1, B_k = q'^-1 for CKKS; BGV (mod_switch.cpp:45-77) carries the plain-modulus factors:
// A_k = p^-1 (p mod t1), m_k = A_k t1, m2_k = t2, B_k = q'^-1 (q' mod t2).  One transform per output limb instead of two and no
// intermediate rows y_k for k < L - 1 in HBM; the limb y_q' that cq needs is produced by the ordinary single drop.
// The second coefficient row arrives through a ring of DEPTH 16-byte loads that follows the order in which the first stage touches
// the registers (0, 16, 2, 18, ...: load_flight's issue order).
template <int LOGN, bool SWAP, bool SCALE2> struct DropPre2A {
    static constexpr bool on = true;
    static constexpr int DEPTH = 8;   // (four spill the same four registers below N = 32768, where the epilogue is the tight spot)
    using G = Geo<LOGN>;
    double q, p_last, p_half, q2_last, q2_half, m, mu, m2, m2u;
    const u64 *row;   // the thread's part of the second coefficient row
    bool odd;
    mutable V2 ring[DEPTH];
    static constexpr int reg_of(int j) { return (j & 1) ? 16 + (j - 1) : j; }
    HP_DEV V2 fetch(int r) const {
        const u64 *a = (G::PB == 0) ? row + ((size_t)(r + (odd ? 1 : 0)) << 10) : row + ((size_t)(r >> G::PB) << 10) + (r & ((1 << G::PB) - 1));
        typedef u64 __attribute__((ext_vector_type(2))) vv;
        const vv v = *reinterpret_cast<const vv *>(a);
        return V2{v.x, v.y};
    }
    HP_DEV void prime(const u64 *crow, u32 tid) {
        odd = (tid & 1u) != 0;
        row = (G::PB == 0) ? crow + (tid & ~1u) : crow + ((size_t)tid << G::PB);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) ring[j] = fetch(reg_of(j));
    }
    HP_DEV void operator()(u64 (&x)[32], int r) const {
        const int j = (r & 14) | (r >> 4);
        V2 c2 = ring[j % DEPTH];
        if (j + DEPTH < 16) ring[j % DEPTH] = fetch(reg_of(j + DEPTH));
        if (SWAP) {
            lazy_swap(x, r);
            const u64 keep = odd ? c2.y : c2.x, send = odd ? c2.x : c2.y;
            const u64 recv = from_pair_lane(send);
            c2 = odd ? V2{recv, keep} : V2{keep, recv};
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double c = from_word(x[r + e]);
            c = (c >= p_half) ? c - p_last : c;
            double d = from_word(e ? c2.y : c2.x);
            d = (d >= q2_half) ? d - q2_last : d;
            if (SCALE2) d = a_modmul(d, m2, m2u, q);
            x[r + e] = U(a_modmul(c, m, mu, q) + d);
        }
    }
};

// ---- passes: the slot schedule of hp_ntt_fast.hip (pass_slots) with the FP64 butterfly ---------------------------------------
template <bool FWD, int S, int S0, int S1, int DEP, class Tab, class Pre = NoPre>
HP_DEV void pass_slots_a(u64 (&x)[32], u64x2 (&ring)[DEP], const Tab &tbl, u32 ncls, u32 cls, double q, const Pre &pre = Pre()) {
    if constexpr (S < S1) {
        constexpr int cnt = 1 << (4 - ilog2c(S + 1));
        constexpr int bit = slot_bit<FWD>(S);
        const u64x2 tw = ring[(S - S0) % DEP];
        if constexpr (S + DEP < S1) ring[(S - S0) % DEP] = tbl.at((u32)(S + DEP), ncls, cls);
        if constexpr (cnt >= 2) {
#pragma unroll
            for (int o = 0; o < cnt; o += 2) {
                const int ra = slot_reg<FWD>(S, o), rb = slot_reg<FWD>(S, o + 1);
                if constexpr (Pre::on && S == 0) {
                    static_assert(!Pre::on || (FWD && S0 == 0), "load-side work: first slot of a forward pass");
                    pre(x, ra);
                    pre(x, ra | bit);
                }
                a_bfly(x[ra], x[ra | bit], D(tw.x), D(tw.y), q);
                a_bfly(x[rb], x[rb | bit], D(tw.x), D(tw.y), q);
                if (o & 2) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (cnt == 2) { if constexpr (S & 1) __builtin_amdgcn_sched_barrier(0); }
            pass_slots_a<FWD, S + 1, S0, S1, DEP, Tab, Pre>(x, ring, tbl, ncls, cls, q, pre);
        } else {
            static_assert(S + 1 < S1, "single-butterfly slots come in pairs");
            const u64x2 tw2 = ring[(S + 1 - S0) % DEP];
            if constexpr (S + 1 + DEP < S1) ring[(S + 1 - S0) % DEP] = tbl.at((u32)(S + 1 + DEP), ncls, cls);
            constexpr int ra = slot_reg<FWD>(S, 0), rb = slot_reg<FWD>(S + 1, 0);
            a_bfly(x[ra], x[ra | bit], D(tw.x), D(tw.y), q);
            a_bfly(x[rb], x[rb | bit], D(tw2.x), D(tw2.y), q);
            if constexpr (((S - 15) & 2) != 0) __builtin_amdgcn_sched_barrier(0);
            pass_slots_a<FWD, S + 2, S0, S1, DEP>(x, ring, tbl, ncls, cls, q);
        }
    }
}
template <bool FWD, int S0, int S1, int DEP, class Tab, class Pre = NoPre>
HP_DEV void run_pass_a(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, double q, const Pre &pre = Pre()) {
    u64x2 ring[DEP];
#pragma unroll
    for (int s = S0; s < S0 + DEP; ++s)
        if (s < S1) ring[(s - S0) % DEP] = tbl.at((u32)s, ncls, cls);
    pass_slots_a<FWD, S0, S0, S1, DEP, Tab, Pre>(x, ring, tbl, ncls, cls, q, pre);
}
template <int BLO, class Tab, class Pre = NoPre>
HP_DEV void fwd_pass_a(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, double q, const Pre &pre = Pre()) {
    run_pass_a<true, 0, (1 << (5 - BLO)) - 1, Tab::depth, Tab, Pre>(x, tbl, ncls, cls, q, pre);
}
template <int BLO, class Tab> HP_DEV void inv_pass_a(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, double q) {
    run_pass_a<false, (1 << BLO) - 1, 31, Tab::depth>(x, tbl, ncls, cls, q);
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
// FLAV (fused drop): 1 CKKS, no addend; 2 CKKS, addend on both polynomials (relinearize's +=, ckks/arith.cpp:70-71); 3 / 4 the
// same with the BGV factors (mod_switch.cpp:70,76); 5 CKKS, addend on polynomial 0 only (rotations, ckks/arith.cpp:75-93);
// 6 / 7 two drops at once (DropPre2A), CKKS / BGV, addend of the first drop on both polynomials
template <int LOGN, bool DROP, int FLAV, bool SPREAD = false, bool WORDS = false>
HP_DEV void ntt_fwd_a_body(const HpNttJob &job, const HpDropArgs *da) {
    using G = Geo<LOGN>;
    using AD = Addr<LOGN, LOGN == 15>;
    __shared__ u32 lds[AD::WORDS];
    __shared__ u64v2 lds_tw[31 * (1 << G::A)];
    TRACE_ENTRY
    const u32 w = hp_xcd_remap(blockIdx.x, job.W);
    HpItem it;
    if (!hp_decode_item(job, w, it)) return;
    const cptr_limba lp = (cptr_limba)(job.limbs_a + __builtin_amdgcn_readfirstlane(it.limb));
    const double q = lp->q, qinv = lp->qinv;
    const bool wide = lp->wide != 0;
    const u32 tid = threadIdx.x;
    AD ad;
    ad.init(tid);
    u64v2 stg = {0, 0};
    if (tid < 31u * (1u << G::A)) stg = ((gptr_u64x2)lp->fwd_k)[tid];
    TRACE_DECL
    TRACE_MARK();   // 0: decoded, staging load issued
    u64 x[32];
    constexpr bool SW = G::PB == 0;   // N = 32768: registers left as loaded, sorted into columns by the lane-pair swap of the first stage
    constexpr bool TWO = FLAV == 6 || FLAV == 7;
    RangeAcc guard;
    DropPre2A<LOGN, SW, FLAV == 7> pre2;
    if constexpr (TWO) pre2.prime(da->comb + (size_t)it.poly * G::N, tid);
    load_flight<LOGN, SW>(it.src, tid, x);
    if (tid < 31u * (1u << G::A)) lds_tw[tid] = stg;
    constexpr bool BGV = FLAV == 3 || FLAV == 4;
#ifdef HP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TRACE_MARK();   // 1: coefficients have arrived
    // pass A: global stages 1..A, wave-uniform twiddles seq[1 .. 2^A - 1]
    if constexpr (TWO) {
        const u32 k = it.limb;
        pre2.q = q; pre2.p_last = D(da->dc.q_last); pre2.p_half = D(da->dc.half_q_last);
        pre2.q2_last = D(da->q2_last); pre2.q2_half = D(da->half_q2_last);
        pre2.m = D(da->dc.t[k]); pre2.mu = D(da->dc.t_h[k]); pre2.m2 = D(da->comb_mul[k]); pre2.m2u = D(da->comb_mul_h[k]);
        fwd_pass_a<G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, q, pre2);
    } else if constexpr (DROP) {
        const u32 k = it.limb;
        const DropPreA<SW, BGV> pre{q, D(da->dc.q_last), D(da->dc.half_q_last), D(da->dc.t[k]), D(da->dc.t_h[k])};
        fwd_pass_a<G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, q, pre);
    } else {
        ConvPre<SW, SPREAD && !WORDS, !SPREAD> pre;
        pre.q = q; pre.qinv = qinv;
        fwd_pass_a<G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, q, pre);
        // N = 32768 reports after the exchange that follows (a divergent branch + two scalar loads here, where it has no SGPR to
        // spare, spilled 60 of them; carried to the end of the kernel the accumulator spilled 6 VGPRs); the smaller degrees report now
        if constexpr (LOGN == 15) guard = pre.acc;
        else if (!SPREAD) pre.acc.report(lp);
    }
    if (wide) a_reduce_all(x, qinv, q);
    TRACE_MARK();   // 2
    exchange<LOGN, LAY_A, LAY_B, true>(x, lds, ad);
    if (!DROP && !SPREAD && LOGN == 15) guard.report(lp);
    TRACE_MARK();   // 3
    // pass B: global stages A+1..A+5, twiddles depend on the 1024-block
    fwd_pass_a<0>(x, LTab(lds_tw), 1u << G::A, tid >> 5, q);
    if (wide) a_reduce_all(x, qinv, q);
    TRACE_MARK();   // 4
    exchange<LOGN, LAY_B, LAY_C, false>(x, lds, ad);
    TRACE_MARK();   // 5
    // pass C: global stages A+6..logN, per-thread twiddles
    fwd_pass_a<0>(x, BTab(lp->fwd_k + 31 * (1 << G::A)), (u32)G::T, tid, q);
    TRACE_MARK();   // 6
    if (!DROP) {
        if constexpr (SPREAD) {
            // digit rows (workspace read by the inner product alone, which takes any word below 2^51 -- level B hands it lazy ones):
            // the centred residue + q, in [q/2, 3q/2], needs no sign fix-up; q and the 2^52 of the conversion are one addend
            // (HP_PACK40 rows: + (q - 1)/2 + 1 instead of + q, hp_device.h)
            const bool p40 = ((job.pack40_mask >> it.limb) & 1u) != 0;
            const double bias = (p40 ? __builtin_floor(q * 0.5) + 1.0 : q) + TWO52;
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = U(a_reduce(D(x[r]), qinv, q) + bias) & 0x000FFFFFFFFFFFFFull;
        } else {
            // canonical residues, as words
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = a_canon(D(x[r]), qinv, q);
        }
    } else if (wide) {
        a_reduce_all(x, qinv, q);
    }
    TRACE_MARK();   // 7: canonical
    exchange<LOGN, LAY_C, LAY_S, false>(x, lds, ad);
    TRACE_MARK();   // 8
    if (!DROP) {
        const size_t off = (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
        if (SPREAD && ((job.pack40_mask >> it.limb) & 1u)) {
            typedef u32 __attribute__((ext_vector_type(2))) v2u;
            u32 *lo = reinterpret_cast<u32 *>(it.dst) + off;
            unsigned short *hi = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(it.dst) + 4 * (size_t)G::N + off);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                __builtin_nontemporal_store(v2u{lo32(x[2 * s]), lo32(x[2 * s + 1])}, reinterpret_cast<v2u *>(lo + ((size_t)s << 7)));
                __builtin_nontemporal_store((unsigned short)((hi32(x[2 * s]) & 0xffu) | ((hi32(x[2 * s + 1]) & 0xffu) << 8)), hi + ((size_t)s << 6));
            }
        } else if (SPREAD && ((job.pack_mask >> it.limb) & 1u)) {
            // HP_PACK48 (hp_device.h): words below 3q/2 of a modulus below 2^47 always fit
            typedef u32 __attribute__((ext_vector_type(2))) v2u;
            u32 *lo = reinterpret_cast<u32 *>(it.dst) + off;
            u32 *hi = reinterpret_cast<u32 *>(it.dst) + G::N + (off >> 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                __builtin_nontemporal_store(v2u{lo32(x[2 * s]), lo32(x[2 * s + 1])}, reinterpret_cast<v2u *>(lo + ((size_t)s << 7)));
                __builtin_nontemporal_store((hi32(x[2 * s]) & 0xffffu) | (hi32(x[2 * s + 1]) << 16), hi + ((size_t)s << 6));
            }
        } else {
            u64 *d = it.dst + off;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                V2 v{x[2 * s], x[2 * s + 1]};
                st_stream(d + ((size_t)s << 7), v);
            }
        }
    } else {
        // rescaling.cpp:72-74 / mod_switch.cpp:72-76 (+ the += of relinearize): ((x - NTT(rem)) * inv) [* (q_last mod t)] [+ addend]
        // as a residue: x and the addend are lazy words of the caller's rows (below 2^51), the result is canonical
        const u32 k = it.limb, p2 = it.poly;
        const u32 voff = ((((tid >> 6)) << 11) + ((tid & 63u) << 1)) << 3;
        const bool has_add = FLAV == 2 || FLAV == 4 || TWO || (FLAV == 5 && __builtin_amdgcn_readfirstlane((p2 & 1u) == 0 ? 1 : 0) != 0);
        const StreamBuf xs(da->x + ((size_t)p2 * da->L + k) * G::N);
        const StreamBuf as(has_add ? da->addend + ((size_t)(p2 >> 1) * da->add_ct_stride + (size_t)(p2 & 1) * da->add_poly_stride + k) * G::N
                                   : da->x);
        const StreamBuf d(da->out + ((size_t)p2 * da->out_stride + k) * G::N);
        const double inv = D(da->dc.inv[k]), invu = D(da->dc.inv_h[k]), ql = D(da->dc.qlt[k]), qlu = D(da->dc.qlt_h[k]);
        RangeAcc acc;
        auto rows = [&](auto add_tag) {
            constexpr bool ADD = decltype(add_tag)::value;
            constexpr int EPI_DEPTH = HP_EPI_DEPTH;
            V2 xr[EPI_DEPTH], ar[EPI_DEPTH];
#pragma unroll
            for (int s = 0; s < EPI_DEPTH; ++s) {
                xr[s] = xs.load(voff, (u32)s << 10);
                if (ADD) ar[s] = as.load(voff, (u32)s << 10);
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const V2 xv = xr[s % EPI_DEPTH];
                V2 av{0, 0};
                if (ADD) av = ar[s % EPI_DEPTH];
                __builtin_amdgcn_sched_barrier(0);
                if (s + EPI_DEPTH < 16) {
                    xr[s % EPI_DEPTH] = xs.load(voff, (u32)(s + EPI_DEPTH) << 10);
                    if (ADD) ar[s % EPI_DEPTH] = as.load(voff, (u32)(s + EPI_DEPTH) << 10);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!TWO) {   // (the two-drop flavours only run inside the fused mult pipelines: their rows are the engine's own, lazy by construction)
                    acc.see(xv.x);
                    acc.see(xv.y);
                    if (ADD) {
                        acc.see(av.x);
                        acc.see(av.y);
                    }
                }
                double v0, v1;
                if constexpr (TWO) {
                    // ((A x + a) - NTT(...)) B: |A x| <= q/2 (1 + 2^-1), a < 2^51, |NTT| < 1.12 * 2^50 (narrow limbs; q/2 for wide ones): below 2^52
                    v0 = a_modmul(a_modmul(from_word(xv.x), inv, invu, q) + from_word(av.x) - D(x[2 * s]), ql, qlu, q);
                    v1 = a_modmul(a_modmul(from_word(xv.y), inv, invu, q) + from_word(av.y) - D(x[2 * s + 1]), ql, qlu, q);
                    d.store(voff + ((u32)s << 10), V2{a_canon(v0, qinv, q), a_canon(v1, qinv, q)});
                    continue;
                }
                v0 = a_modmul(from_word(xv.x) - D(x[2 * s]), inv, invu, q);
                v1 = a_modmul(from_word(xv.y) - D(x[2 * s + 1]), inv, invu, q);
                if (BGV) {
                    v0 = a_modmul(v0, ql, qlu, q);
                    v1 = a_modmul(v1, ql, qlu, q);
                }
                if (ADD) {
                    v0 += from_word(av.x);
                    v1 += from_word(av.y);
                }
                d.store(voff + ((u32)s << 10), V2{a_canon(v0, qinv, q), a_canon(v1, qinv, q)});
            }
        };
        if constexpr (FLAV == 2 || FLAV == 4 || TWO) rows(std::true_type{});
        else if constexpr (FLAV == 1 || FLAV == 3) rows(std::false_type{});
        else if (has_add) rows(std::true_type{});
        else rows(std::false_type{});
        if (!TWO) acc.report(lp);
    }
    TRACE_MARK();   // 9: stores issued
    TRACE_FLUSH();
}

template <int LOGN, bool SPREAD>
__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW) k_ntt_fwd_a(HpNttJob job) {
    ntt_fwd_a_body<LOGN, false, 0, SPREAD>(job, nullptr);
}
// the digit spread from WORDS (limb-range stage of the limb-sharded mode: coefficient rows that came from other ranks)
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW) k_ntt_fwd_a_words(HpNttJob job) {
    ntt_fwd_a_body<LOGN, false, 0, true, true>(job, nullptr);
}
template <int LOGN, int FLAV>
__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW) k_ntt_fwd_drop_a(HpNttJob job, HpDropArgs da) {
    ntt_fwd_a_body<LOGN, true, FLAV>(job, &da);
}

// ---- inverse (always returns canonical residues: intt_negacyclic_inplace of ntt.h:88-92) -----------------------------------
template <int LOGN> struct InvGeoA {
    static constexpr int LPW = LOGN >= 14 ? 1 : (512 >> (LOGN - 5));   // limbs of one modulus per workgroup (hp_ntt_fast.hip: InvGeo)
    static constexpr int TT = Geo<LOGN>::T * LPW;
    static constexpr bool STREAM_EPILOGUE = LOGN <= 13;
};

// MIX (the second coefficient row of drop_two_last_a in ONE launch): the coefficients modulo q' of y = A x + a - NTT(K cp), the row the
// first drop would leave in the limb of q', are INTT(A x + a) - K cp by linearity: the input is combined from the row x and the addend
// row a while loading, K times the centred coefficients of the previous drop is subtracted from the output
template <int LOGN, bool PSCAL, bool MIX>
HP_DEV void ntt_inv_a_body(const HpNttJob &job, const HpInvMixArgs *mx) {
    using G = Geo<LOGN>;
    constexpr int LPW = InvGeoA<LOGN>::LPW, TT = InvGeoA<LOGN>::TT;
    __shared__ u32 lds_all[Addr<LOGN>::WORDS * LPW];
    __shared__ u64v2 lds_tw[31 * 32];
    const u32 sub = threadIdx.x / G::T, tid = threadIdx.x % G::T;
    u32 *lds = lds_all + sub * Addr<LOGN>::WORDS;
    HpItem it;
    bool active = true;
    if (LPW == 1) {
        const u32 w = hp_xcd_remap(blockIdx.x, job.W);
        const u32 k = w / job.P, p = w % job.P;
        it.src = job.src + ((size_t)p * job.src_pstride + (size_t)k * job.src_kstride) * G::N;
        it.dst = job.dst + ((size_t)p * job.dst_pstride + k) * G::N;
        it.limb = k;
        it.poly = p;
    } else {
        const u32 bpm = (job.P + LPW - 1) / LPW;
        const u32 wb = hp_xcd_remap(blockIdx.x, job.L * bpm);
        const u32 k = wb / bpm, p0 = (wb % bpm) * LPW + sub;
        active = p0 < job.P;
        const u32 p = active ? p0 : job.P - 1;
        it.src = job.src + ((size_t)p * job.src_pstride + (size_t)k * job.src_kstride) * G::N;
        it.dst = job.dst + ((size_t)p * job.dst_pstride + k) * G::N;
        it.limb = k;
        it.poly = p;
    }
    const cptr_limba lp = (cptr_limba)(job.limbs_a + __builtin_amdgcn_readfirstlane(it.limb));
    const double q = lp->q, qinv = lp->qinv;
    const bool wide = lp->wide != 0;
    Addr<LOGN> ad;
    ad.init(tid);
    constexpr int NSTG = (31 * 32 + TT - 1) / TT;
    u64v2 stg[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const u32 e = threadIdx.x + (u32)i * TT;
        stg[i] = (e < 31u * 32u) ? ((gptr_u64x2)(lp->inv_k + 31))[e] : u64v2{0, 0};
    }
    u64 x[32];
    {
        const u64 *s = it.src + (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const V2 v = ld_stream(s + ((size_t)r << 7));
            x[2 * r] = v.x;
            x[2 * r + 1] = v.y;
        }
    }
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const u32 e = threadIdx.x + (u32)i * TT;
        if (e < 31u * 32u) lds_tw[e] = stg[i];
    }
    if (!MIX) {   // (MIX rows are the engine's own)
        RangeAcc acc;
#pragma unroll
        for (int r = 0; r < 32; ++r) acc.see(x[r]);
        acc.report(lp);
    }
    if constexpr (MIX) {
        const u64 *as = mx->add + ((size_t)(it.poly >> 1) * mx->add_ct_stride + (size_t)(it.poly & 1) * mx->add_poly_stride) * G::N +
                        (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
        const double A = D(mx->A), Au = D(mx->A_h);
        V2 ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = ld_stream(as + ((size_t)r << 7));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const V2 av = ar[r % 4];
            if (r + 4 < 16) ar[r % 4] = ld_stream(as + ((size_t)(r + 4) << 7));
            // |A x| <= q (1/2 + 2^-1), a < 2^51: narrow limbs take lazy words below 2^45 (sum < 2^50), wide ones are reduced right below
            x[2 * r] = U(a_modmul(from_word(x[2 * r]), A, Au, q) + from_word(av.x));
            x[2 * r + 1] = U(a_modmul(from_word(x[2 * r + 1]), A, Au, q) + from_word(av.y));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = U(from_word(x[r]));
    }
    if (wide) a_reduce_all(x, qinv, q);   // lazy words up to 2 q
    exchange<LOGN, LAY_S, LAY_C, false>(x, lds, ad);
    inv_pass_a<0>(x, STab(lp->inv_k), 1u, 0u, q);
    if (wide) a_reduce_all(x, qinv, q);
    exchange<LOGN, LAY_C, LAY_B, false>(x, lds, ad);
    __syncthreads();   // the staged twiddles are read by other waves from here on
    inv_pass_a<0>(x, LTab(lds_tw), 32u, tid & 31u, q);
    if (wide) a_reduce_all(x, qinv, q);
    exchange<LOGN, LAY_B, LAY_A, true>(x, lds, ad);
    inv_pass_a<G::PB>(x, BTab(lp->inv_k + 31 + 31 * 32), (u32)G::T, tid, q);
    if (wide) a_reduce_all(x, qinv, q);
    const double psc = D(job.post_scalar), psu = D(job.post_scalar_h);
    // rows for a caller: words; rows for the digit-spread launch of the same key switch (job.dst_f64): the doubles themselves
    const double wbias = job.dst_f64 ? 0.0 : TWO52;
    const u64 wmask = job.dst_f64 ? ~0ull : 0x000FFFFFFFFFFFFFull;
    // MIX: v - K centre(c), back into (-q, q) unless the post-scalar multiplication does that anyway
    const u64 *cprow = MIX ? mx->cprev + (size_t)it.poly * G::N : nullptr;
    const double mK = MIX ? D(mx->K) : 0.0, mKu = MIX ? D(mx->K_h) : 0.0, mpq = MIX ? D(mx->prev_q) : 0.0, mph = MIX ? D(mx->prev_half) : 0.0;
    auto mix = [&](double v, u64 cw) {
        double c = from_word(cw);
        c = (c >= mph) ? c - mpq : c;
        v -= a_modmul(c, mK, mKu, q);
        return PSCAL ? v : a_reduce(v, qinv, q);
    };
    if constexpr (InvGeoA<LOGN>::STREAM_EPILOGUE) {
        // N <= 8192: transpose once more so that the psi^-i N^-1 pairs are read and the words written 16 contiguous bytes per lane
        exchange<LOGN, LAY_A, LAY_S, true>(x, lds, ad);
        if (!active) return;
        const size_t off = (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
        const BTab sc(lp->inv_ref + G::N);
        u64 *d = it.dst + off;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 2) {
            u64x2 f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = sc.at((u32)(s0 + (e >> 1)), 128u, (u32)off + (u32)(e & 1));
            V2 cw[2] = {{0, 0}, {0, 0}};
            if constexpr (MIX) {
                cw[0] = ld_stream(cprow + off + ((size_t)s0 << 7));
                cw[1] = ld_stream(cprow + off + ((size_t)(s0 + 1) << 7));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 2 * s0 + e;
                double v = a_modmul(D(x[r]), D(f[e].x), D(f[e].y), q);      // ntt.cpp:214-222 as a residue
                if constexpr (MIX) v = mix(v, (e & 1) ? cw[e >> 1].y : cw[e >> 1].x);
                if (PSCAL) v = a_modmul(v, psc, psu, q);                    // mod_switch.cpp:49
                x[r] = U(a_nonneg(v, q) + wbias) & wmask;
            }
            st_stream(d + ((size_t)s0 << 7), V2{x[2 * s0], x[2 * s0 + 1]});
            st_stream(d + ((size_t)(s0 + 1) << 7), V2{x[2 * s0 + 2], x[2 * s0 + 3]});
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    {
        const BTab sc(lp->inv_ref + G::N);
        u64 *d = it.dst + ((size_t)tid << G::PB);
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 4) {
            u64x2 f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e, kk = r >> G::PB, pp = r & ((1 << G::PB) - 1);
                f[e] = sc.at((u32)kk, 1024u, (tid << G::PB) + (u32)pp);
            }
            u64 cw[4] = {0, 0, 0, 0};
            if constexpr (MIX) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = r0 + e, kk = r >> G::PB, pp = r & ((1 << G::PB) - 1);
                    cw[e] = cprow[((size_t)kk << 10) + ((size_t)tid << G::PB) + pp];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                double v = a_modmul(D(x[r]), D(f[e].x), D(f[e].y), q);
                if constexpr (MIX) v = mix(v, cw[e]);
                if (PSCAL) v = a_modmul(v, psc, psu, q);
                x[r] = U(a_nonneg(v, q) + wbias) & wmask;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!active) return;
        if (G::PB == 0) {
            const bool odd = (tid & 1u) != 0;
            u64 *dp = it.dst + (tid & ~1u);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const u64 recv = from_pair_lane(odd ? x[2 * p] : x[2 * p + 1]);
                V2 v;
                v.x = odd ? recv : x[2 * p];
                v.y = odd ? x[2 * p + 1] : recv;
                st_stream(dp + ((size_t)(2 * p + (odd ? 1 : 0)) << 10), v);
            }
        }
#pragma unroll
        for (int kk = 0; kk < (G::PB == 0 ? 0 : (1 << G::A)); ++kk) {
            if (G::PB == 0) {
            } else {
#pragma unroll
                for (int pp = 0; pp < (1 << G::PB); pp += 2) {
                    V2 v{x[(kk << G::PB) | pp], x[(kk << G::PB) | pp | 1]};
                    st_stream(d + ((size_t)kk << 10) + pp, v);
                }
            }
        }
    }
}

template <int LOGN, bool PSCAL>
__global__ void __launch_bounds__(InvGeoA<LOGN>::TT, Geo<LOGN>::MINW) k_ntt_inv_a(HpNttJob job) {
    ntt_inv_a_body<LOGN, PSCAL, false>(job, nullptr);
}
template <int LOGN, bool PSCAL>
__global__ void __launch_bounds__(InvGeoA<LOGN>::TT, Geo<LOGN>::MINW) k_ntt_inv_mix_a(HpNttJob job, HpInvMixArgs mx) {
    ntt_inv_a_body<LOGN, PSCAL, true>(job, &mx);
}

template <int LOGN> hipError_t launch_inv_mix_a(const HpNttJob &job, const HpInvMixArgs &mx, hipStream_t stream) {
    constexpr int LPW = InvGeoA<LOGN>::LPW, TT = InvGeoA<LOGN>::TT;
    if (!job.inverse || job.mode != HP_NTT_BATCH || job.pair_moduli || job.L != 1) return hipErrorNotSupported;
    const u32 grid = LPW == 1 ? job.W : job.L * ((job.P + LPW - 1) / LPW);
    if (job.use_post_scalar) k_ntt_inv_mix_a<LOGN, true><<<grid, TT, 0, stream>>>(job, mx);
    else k_ntt_inv_mix_a<LOGN, false><<<grid, TT, 0, stream>>>(job, mx);
    return hipGetLastError();
}

template <int LOGN> hipError_t launch_a(const HpNttJob &job, hipStream_t stream) {
    if (!job.inverse) {
        // (HP_NTT_HKS: the lifted digits of the hybrid key switch, in place -- canonical residues in, canonical residues out: the plain kernel)
        if (job.mode != HP_NTT_BATCH && job.mode != HP_NTT_SPREAD && job.mode != HP_NTT_HKS) return hipErrorNotSupported;
        if (job.mode == HP_NTT_SPREAD && job.src_words) k_ntt_fwd_a_words<LOGN><<<job.W, Geo<LOGN>::T, 0, stream>>>(job);
        else if (job.mode == HP_NTT_SPREAD) k_ntt_fwd_a<LOGN, true><<<job.W, Geo<LOGN>::T, 0, stream>>>(job);
        else k_ntt_fwd_a<LOGN, false><<<job.W, Geo<LOGN>::T, 0, stream>>>(job);
        return hipGetLastError();
    }
    constexpr int LPW = InvGeoA<LOGN>::LPW, TT = InvGeoA<LOGN>::TT;
    if (job.mode != HP_NTT_BATCH || job.pair_moduli) return hipErrorNotSupported;
    const u32 grid = LPW == 1 ? job.W : job.L * ((job.P + LPW - 1) / LPW);
    if (job.use_post_scalar) k_ntt_inv_a<LOGN, true><<<grid, TT, 0, stream>>>(job);
    else k_ntt_inv_a<LOGN, false><<<grid, TT, 0, stream>>>(job);
    return hipGetLastError();
}

template <int LOGN> hipError_t launch_drop_a(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream) {
    if (da.fin_on || da.raw_input) return hipErrorNotSupported;
    int flav = 0;
    if (da.comb) flav = (da.addend && da.add_mask == 3u) ? (da.dc.bgv ? 7 : 6) : 0;   // two drops at once
    else if (!da.addend || da.add_mask == 0) flav = 1;
    else if (da.add_mask == 3u) flav = 2;
    else if (da.add_mask == 1u && !da.dc.bgv) flav = 5;
    if (flav && flav < 5 && da.dc.bgv) flav += 2;
#define HP_DROP_A(F) k_ntt_fwd_drop_a<LOGN, F><<<job.W, Geo<LOGN>::T, 0, stream>>>(job, da)
    if (flav == 1) HP_DROP_A(1);
    else if (flav == 2) HP_DROP_A(2);
    else if (flav == 3) HP_DROP_A(3);
    else if (flav == 4) HP_DROP_A(4);
    else if (flav == 5) HP_DROP_A(5);
    else if (flav == 6) HP_DROP_A(6);
    else if (flav == 7) HP_DROP_A(7);
    else return hipErrorNotSupported;
#undef HP_DROP_A
    return hipGetLastError();
}

} // namespace

#ifdef HP_TRACE
extern "C" int hp_debug_trace_a(u64 *out, size_t words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), words * sizeof(u64));
}
#endif

hipError_t hp_launch_ntt_a(const HpNttJob &job, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    if (!job.limbs_a) return hipErrorInvalidValue;
    switch (job.logn) {
    case 11: return launch_a<11>(job, stream);
    case 12: return launch_a<12>(job, stream);
    case 13: return launch_a<13>(job, stream);
    case 14: return launch_a<14>(job, stream);
    case 15: return launch_a<15>(job, stream);
    default: return hipErrorNotSupported;
    }
}

hipError_t hp_launch_ntt_a_inv_mix(const HpNttJob &job, const HpInvMixArgs &mx, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    if (!job.limbs_a) return hipErrorInvalidValue;
    switch (job.logn) {
    case 11: return launch_inv_mix_a<11>(job, mx, stream);
    case 12: return launch_inv_mix_a<12>(job, mx, stream);
    case 13: return launch_inv_mix_a<13>(job, mx, stream);
    case 14: return launch_inv_mix_a<14>(job, mx, stream);
    case 15: return launch_inv_mix_a<15>(job, mx, stream);
    default: return hipErrorNotSupported;
    }
}

hipError_t hp_launch_ntt_a_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    if (!job.limbs_a) return hipErrorInvalidValue;
    switch (job.logn) {
    case 11: return launch_drop_a<11>(job, da, stream);
    case 12: return launch_drop_a<12>(job, da, stream);
    case 13: return launch_drop_a<13>(job, da, stream);
    case 14: return launch_drop_a<14>(job, da, stream);
    case 15: return launch_drop_a<15>(job, da, stream);
    default: return hipErrorNotSupported;
    }
}
