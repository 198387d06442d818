// hp_ntt_fast.hip -- register-tiled negacyclic NTT / INTT for N = 2^11 .. 2^15 on gfx950.
//
// One workgroup transforms one RNS limb.  The limb is read from HBM once and written once
// (algorithmic traffic 16*N bytes); everything in between happens in
// registers and LDS:
//
//   * every thread keeps 32 coefficients (64 VGPRs) and runs up to five radix-2 stages on
//     them per pass -- the butterflies are exactly the reference's lazy Harvey butterflies
//     (ntt.cpp:160-166), only their schedule changes, so raw output words are identical;
//   * N = 32 * T coefficients, T = N/32 threads; logN = a + 5 + 5 stages = three passes
//     (A: top a bits, B: bits 9..5, C: bits 4..0 of the coefficient index; the inverse runs
//     them in the opposite order, C' B' A');
//   * between passes the workgroup transposes through LDS, 32 bits at a time (low words, then
//     high words), so a whole limb needs only 4*N bytes of LDS: 128 KiB at N = 32768 (one
//     workgroup of 16 waves per CU) and 64 KiB at N = 16384 (two workgroups per CU);
//   * the exchange between the passes that work inside a 1024-coefficient block (B <-> C) and
//     the transposition that makes the HBM stores/loads of the contiguous pass coalesced are
//     wave-local: each wave owns 2048 consecutive coefficients there, so they need no
//     s_barrier and waves drift apart, overlapping one wave's LDS traffic with another's
//     integer multiplies.  Only the A <-> B exchange is workgroup-wide (3 barriers);
//   * LDS word address of coefficient i is i ^ ((i >> 5) & 31) (forward kernels at N = 32768: i + (i >> 5), a padded buffer
//     whose register offsets are all immediates): every ds_read_b32 / ds_write_b32 of every exchange hits 32 distinct banks
//     per 32-lane group;
//   * twiddles are (w, floor(w*2^64/q)) pairs read as one 16-byte load from per-modulus tables
//     laid out [slot][class] (hp_tables.cpp) so a wavefront reads consecutive pairs; the
//     tables are shared by the whole batch and live in L2 -- work items are numbered
//     modulus-major and handed to XCDs in contiguous slices (hp_xcd_remap) to keep them there.
//
// The kernel is bound by the integer ALUs, not by HBM: 10 multiply-type instructions of 16 per butterfly
// (tools/ubench/*.hip), 88 % VALUBusy in the PMC passes at 35-44 % of HBM peak; see DESIGN.md 4.1.
#include "hp_kernels.h"
#include "hp_ntt_job.h"
#include <type_traits>

#include "hp_ntt_tile.h"

namespace {

// One slot (or, in the last stage of a pass where every butterfly has its own twiddle, two slots) of a pass;
// recursion over the slot number keeps every register index a compile-time constant.
// Pre (first pass of a forward kernel): work that belongs to the LOADS is done here, per register pair (r, r + 1) = one 16-byte
// load, right before the first butterfly that touches it -- so the first stage runs while the later loads are still in flight
// instead of after all sixteen:
//   SwapPre  N = 32768: registers still hold the loads as they arrived; the lane-pair swap sorts them into columns (load_flight)
//   DropPre  fused drop-last-prime: that swap, then the Barrett / centring / [* t] prologue of rescaling.cpp:54-69, mod_switch.cpp:52-70
template <bool SWAP, bool BGV, bool SMALL> struct DropPre {
    static constexpr bool on = true;
    u64 q, bc, bump, half, tk, tkh;
    u32 n0, n1;
    HP_DEV void operator()(u64 (&x)[32], int r) const {
        if (SWAP) lazy_swap(x, r);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const u64 c = x[r + e];
            u64 v = SMALL ? hp_strict(c, q) : hp_strict(hp_barrett_lazy_nq(c, bc, n0, n1), q);   // SMALL: c < q_last <= 2q
            if (c >= half) v += bump;
            if (BGV) v = hp_harvey_lazy_nq(v, tk, tkh, n0, n1);
            x[r + e] = v;
        }
    }
};

template <bool FWD, int S, int S0, int S1, int D, class Tab, class Pre = NoPre>
HP_DEV void pass_slots(u64 (&x)[32], u64x2 (&ring)[D], const Tab &tbl, u32 ncls, u32 cls, u64 two_q, u32 n0, u32 n1, const Pre &pre = Pre()) {
    if constexpr (S < S1) {
        constexpr int cnt = 1 << (4 - ilog2c(S + 1));   // butterflies that use this slot's twiddle
        constexpr int bit = slot_bit<FWD>(S);
        const u64x2 tw = ring[(S - S0) % D];
        if constexpr (S + D < S1) ring[(S - S0) % D] = tbl.at((u32)(S + D), ncls, cls);
        if constexpr (cnt >= 2) {
#pragma unroll
            for (int o = 0; o < cnt; o += 2) {
                const int ra = slot_reg<FWD>(S, o), rb = slot_reg<FWD>(S, o + 1);
                if constexpr (Pre::on && S == 0) {
                    static_assert(!Pre::on || (FWD && S0 == 0), "load-side work: first slot of a forward pass");
                    pre(x, ra);          // rb == ra + 1: one 16-byte load
                    pre(x, ra | bit);
                }
                if constexpr (Tab::scalar) hp_butterfly2_nq_sw(x[ra], x[ra | bit], x[rb], x[rb | bit], tw.x, tw.y, tw.x, tw.y, two_q, n0, n1);
                else hp_butterfly2_nq(x[ra], x[ra | bit], x[rb], x[rb | bit], tw.x, tw.y, tw.x, tw.y, two_q, n0, n1);
                if (o & 2) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (cnt == 2) { if constexpr (S & 1) __builtin_amdgcn_sched_barrier(0); }
            pass_slots<FWD, S + 1, S0, S1, D, Tab, Pre>(x, ring, tbl, ncls, cls, two_q, n0, n1, pre);
        } else {
            static_assert(S + 1 < S1, "single-butterfly slots come in pairs");
            const u64x2 tw2 = ring[(S + 1 - S0) % D];
            if constexpr (S + 1 + D < S1) ring[(S + 1 - S0) % D] = tbl.at((u32)(S + 1 + D), ncls, cls);
            constexpr int ra = slot_reg<FWD>(S, 0), rb = slot_reg<FWD>(S + 1, 0);
            if constexpr (Tab::scalar) hp_butterfly2_nq_sw(x[ra], x[ra | bit], x[rb], x[rb | bit], tw.x, tw.y, tw2.x, tw2.y, two_q, n0, n1);
            else hp_butterfly2_nq(x[ra], x[ra | bit], x[rb], x[rb | bit], tw.x, tw.y, tw2.x, tw2.y, two_q, n0, n1);
            if constexpr (((S - 15) & 2) != 0) __builtin_amdgcn_sched_barrier(0);
            pass_slots<FWD, S + 2, S0, S1, D>(x, ring, tbl, ncls, cls, two_q, n0, n1);
        }
    }
}

template <bool FWD, int S0, int S1, int D, class Tab, class Pre = NoPre>
HP_DEV void run_pass(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, u64 nq, u64 two_q, const Pre &pre = Pre()) {
    u64x2 ring[D];
#pragma unroll
    for (int s = S0; s < S0 + D; ++s)
        if (s < S1) ring[(s - S0) % D] = tbl.at((u32)s, ncls, cls);
    pass_slots<FWD, S0, S0, S1, D, Tab, Pre>(x, ring, tbl, ncls, cls, two_q, (u32)nq, (u32)(nq >> 32), pre);
}

// forward: stages on register bits BHI..BLO (descending)
template <int BHI, int BLO, class Tab, class Pre = NoPre>
HP_DEV void fwd_pass(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, u64 nq, u64 two_q, const Pre &pre = Pre()) {
    static_assert(BHI == 4, "forward passes start at register bit 4");
    run_pass<true, 0, (1 << (5 - BLO)) - 1, Tab::depth, Tab, Pre>(x, tbl, ncls, cls, nq, two_q, pre);
}

// inverse: stages on register bits BLO..BHI (ascending)
template <int BLO, int BHI, class Tab, int D = Tab::depth>
HP_DEV void inv_pass(u64 (&x)[32], const Tab tbl, u32 ncls, u32 cls, u64 nq, u64 two_q) {
    static_assert(BHI == 4, "inverse passes end at register bit 4");
    run_pass<false, (1 << BLO) - 1, 31, D>(x, tbl, ncls, cls, nq, two_q);
}

#ifdef HP_TRACE
__device__ u64 g_trace[2 * 2048 * 16 * HP_TRACE_SLOTS];
#endif

// FLAV (fused drop only): 0 = every option decided at run time; 1..5 = the shapes of the CKKS / BGV pipelines with the options
// fixed at compile time (Barrett prologue, no final multiplication; 1: CKKS, no addend; 2: CKKS, addend on both polynomials;
// 3 / 4: the same with the BGV factors; 5: CKKS, addend on polynomial 0 only = rotations), which takes ~100 uniform branches
// out of the prologue and the store loop
template <int LOGN, bool DROP, int FLAV = 0, bool SMALL = false>
HP_DEV void ntt_fwd_body(const HpNttJob &job, const HpDropArgs *da) {
    using G = Geo<LOGN>;
    using AD = Addr<LOGN, LOGN == 15>;   // (re-checked in round 3 with the general PB form below: N = 16384 / 8192 still spill 28 / 39-46 registers with it)
    __shared__ u32 lds[AD::WORDS];
    __shared__ u64v2 lds_tw[31 * (1 << G::A)];
    TRACE_ENTRY
    const u32 w = hp_xcd_remap(blockIdx.x, job.W);
    HpItem it;
    if (!hp_decode_item(job, w, it)) return;
    // the limb's constants and table pointers through the scalar cache (constant address space): as vector loads they would
    // queue behind the coefficient loads and the first pass could not start before nearly all of those are back
    const cptr_limb lp = (cptr_limb)(job.limbs + __builtin_amdgcn_readfirstlane(it.limb));
    const u64 q = lp->q, two_q = lp->two_q, nq = lp->neg_q;
    const u32 tid = threadIdx.x;
    AD ad;
    ad.init(tid);
    // stage the middle pass's twiddles (one pair per thread): the load is issued first, the LDS write after
    // the coefficient loads are in flight; it becomes visible through the barriers of the A->B exchange
    u64v2 stg = {0, 0};
    if (tid < 31u * (1u << G::A)) stg = ((gptr_u64x2)lp->fwd_k)[tid];

    TRACE_DECL
    TRACE_MARK();
    u64 x[32];
    // load-side work deferred into the first stage: the plain N = 32768 kernel (lane-pair swap) and the compile-time flavours of
    // the fused drop (swap + Barrett / centring prologue); the run-time flavour 0 (hybrid key switch inputs) keeps the eager form
    // (N = 32768 only: -2 % on the launch there, +2 % at N = 8192 where two workgroups per CU hide the load phase anyway)
    constexpr bool LZ_DROP = DROP && FLAV != 0 && G::PB == 0;
    constexpr bool LZ = (!DROP && G::PB == 0) || LZ_DROP;   // registers left as loaded
    load_flight<LOGN, LZ>(it.src, tid, x);
    if (tid < 31u * (1u << G::A)) lds_tw[tid] = stg;
    if (DROP) {
        // rescaling.cpp:54-69 / mod_switch.cpp:52-70 while the coefficients are still in flight order:
        // rem = strict_barrett_{q_k}(c) (+ q_k - (q_last mod q_k) if c >= q_last/2) (BGV: * t)
        const u32 k = it.limb;
        const u64 bc = lp->barrett_c, bump = q - da->dc.r[k], half = da->dc.half_q_last;
        const u64 tk = da->dc.t[k], tkh = da->dc.t_h[k];
        const bool bgv = FLAV ? (FLAV == 3 || FLAV == 4) : da->dc.bgv != 0;
        if (!LZ_DROP && (FLAV || !da->raw_input)) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const u64 c = x[r];
                u64 v = SMALL ? hp_strict(c, q) : hp_strict(hp_barrett_lazy_nq(c, bc, (u32)nq, (u32)(nq >> 32)), q);
                if (c >= half) v += bump;
                if (bgv) v = hp_harvey_lazy_nq(v, tk, tkh, (u32)nq, (u32)(nq >> 32));
                x[r] = v;
            }
        }
        if (!FLAV && da->comb) {
            // hybrid key switch, ModDown merged with the rescale: x = rem_k + (P mod q_k) * centre_k(c), c = the strict
            // coefficients modulo q_last of the relinearised last limb (one row per polynomial, read by every limb)
            // Eight registers (four 16-byte loads) at a time with the next four loads in flight.  The loads are inline asm: left
            // to itself the compiler, short of registers beside x, issues the sixteen loads one by one with a full wait after
            // each.  It does not count asm loads, so the waits are explicit (vector loads return in order: "at most four
            // outstanding" means the current four are back); each wait is tied to the registers it guards.
            typedef u64 __attribute__((ext_vector_type(2))) vv;
            const u64 *crow = da->comb + (size_t)it.poly * G::N;
            const u64 pm = da->comb_mul[k], pmh = da->comb_mul_h[k], cbump = q - da->comb_r[k], chalf = da->comb_half;
            const bool odd = (tid & 1u) != 0;
            vv ca[4], cb[4];
            auto issue = [&](vv (&d)[4], int ch) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 8 * ch + 2 * i;
                    const u64 *a = (G::PB == 0) ? crow + (tid & ~1u) + ((size_t)(r + (odd ? 1 : 0)) << 10)
                                                : crow + ((size_t)tid << G::PB) + ((size_t)(r >> G::PB) << 10) + (r & ((1 << G::PB) - 1));
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d[i]) : "v"(a) : "memory");
                }
            };
            issue(ca, 0);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                vv(&cur)[4] = (ch & 1) ? cb : ca;
                vv(&nxt)[4] = (ch & 1) ? ca : cb;
                if (ch + 1 < 4) {
                    issue(nxt, ch + 1);
                    asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u64 c0 = cur[i].x, c1 = cur[i].y;
                    if (G::PB == 0) {   // lane-pair swap as in load_flight
                        const u64 keep = odd ? c1 : c0, send = odd ? c0 : c1;
                        const u64 recv = from_pair_lane(send);
                        c0 = odd ? recv : keep;
                        c1 = odd ? keep : recv;
                    }
                    const int r = 8 * ch + 2 * i;
                    u64 v0 = hp_strict(hp_barrett_lazy_nq(c0, bc, (u32)nq, (u32)(nq >> 32)), q);
                    u64 v1 = hp_strict(hp_barrett_lazy_nq(c1, bc, (u32)nq, (u32)(nq >> 32)), q);
                    if (c0 >= chalf) v0 += cbump;
                    if (c1 >= chalf) v1 += cbump;
                    x[r] = hp_add_lazy(x[r], hp_harvey_lazy_nq(v0, pm, pmh, (u32)nq, (u32)(nq >> 32)), two_q);
                    x[r + 1] = hp_add_lazy(x[r + 1], hp_harvey_lazy_nq(v1, pm, pmh, (u32)nq, (u32)(nq >> 32)), two_q);
                }
            }
        }
    }
#ifdef HP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TRACE_MARK();   // 1: coefficients have arrived
    // pass A: global stages 1..A, wave-uniform twiddles seq[1 .. 2^A - 1]
    if constexpr (LZ_DROP) {
        const u32 k = it.limb;
        const DropPre<G::PB == 0, (FLAV == 3 || FLAV == 4), SMALL> pre{q, lp->barrett_c, q - da->dc.r[k], da->dc.half_q_last, da->dc.t[k], da->dc.t_h[k],
                                                    (u32)nq, (u32)(nq >> 32)};
        fwd_pass<4, G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, nq, two_q, pre);
    } else if constexpr (LZ) {
        fwd_pass<4, G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, nq, two_q, SwapPre());
    } else {
        fwd_pass<4, G::PB, STab>(x, STab(lp->fwd_ref + 1), 1u, 0u, nq, two_q);
    }
    TRACE_MARK();   // 2
    exchange<LOGN, LAY_A, LAY_B, true>(x, lds, ad);
    TRACE_MARK();   // 3
    // pass B: global stages A+1..A+5, twiddles depend on the 1024-block
    fwd_pass<4, 0>(x, LTab(lds_tw), 1u << G::A, tid >> 5, nq, two_q);
    TRACE_MARK();   // 4
    exchange<LOGN, LAY_B, LAY_C, false>(x, lds, ad);
    TRACE_MARK();   // 5
    // pass C: global stages A+6..logN, per-thread twiddles
    fwd_pass<4, 0>(x, BTab(lp->fwd_k + 31 * (1 << G::A)), (u32)G::T, tid, nq, two_q);
    TRACE_MARK();   // 6
    // final fold (ntt.cpp:171-175)
    {
        const u32 k = lp->k, fix = lp->fix;
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = hp_shift_fold(x[r], q, k, fix);
    }
    TRACE_MARK();   // 7: fold done
    exchange<LOGN, LAY_C, LAY_S, false>(x, lds, ad);
    TRACE_MARK();   // 8
    // store, layout S: 16 bytes per lane, a wave writes 1 KiB of consecutive words per instruction
    if (!DROP) {
        const size_t off = (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
        if (job.mode == HP_NTT_SPREAD && ((job.pack_mask >> it.limb) & 1u)) {
            // HP_PACK48 (hp_device.h): low words as 8 bytes per lane, high 16 bits of the two words as 4 bytes per lane
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
        // rescaling.cpp:72-74 / mod_switch.cpp:72-76 (+ the += of relinearize, ckks/arith.cpp:70-71):
        // out = ((x - NTT(rem)) * inv) [* (q_last mod t)] [+ addend], all in the lazy representation
        const u32 k = it.limb, p2 = it.poly;
        // the three streams (x row, addend row, output row) through buffer descriptors: one lane offset for all of them, the row
        // step (s << 7 words) in an SGPR for the loads, in the instruction's immediate / one 32-bit add for the stores -- no 64-bit
        // address arithmetic on the vector ALU in the epilogue
        const u32 voff = ((((tid >> 6)) << 11) + ((tid & 63u) << 1)) << 3;
        // (wave-uniform; said explicitly so that the run-time flavour branches on an SGPR instead of masking lanes)
        // flavour 5 (rotations / conjugations, ckks/arith.cpp:75-93): the moved c0 is added to polynomial 0 only
        const bool has_add = FLAV == 2 || FLAV == 4 || (FLAV == 5 && __builtin_amdgcn_readfirstlane((p2 & 1u) == 0 ? 1 : 0) != 0) ||
                             (FLAV == 0 && __builtin_amdgcn_readfirstlane((da->addend && ((da->add_mask >> (p2 & 1)) & 1u)) ? 1 : 0) != 0);
        const StreamBuf xs(da->x + ((size_t)p2 * da->L + k) * G::N);
        const StreamBuf as(has_add ? da->addend + ((size_t)(p2 >> 1) * da->add_ct_stride + (size_t)(p2 & 1) * da->add_poly_stride + k) * G::N
                                   : da->x);
        const StreamBuf d(da->out + ((size_t)p2 * da->out_stride + k) * G::N);
        const u64 inv = da->dc.inv[k], invh = da->dc.inv_h[k], ql = da->dc.qlt[k], qlh = da->dc.qlt_h[k];
        const bool bgv = FLAV ? (FLAV == 3 || FLAV == 4) : da->dc.bgv != 0, fin_on = FLAV ? false : da->fin_on != 0;
        const u64 fin = da->fin[k], finh = da->fin_h[k];
        const u32 n0 = (u32)nq, n1 = (u32)(nq >> 32);
        // one copy of the row loop per value of has_add where it is only known at run time (flavours 0 and 5): the loop body
        // then has no branch on it
        auto rows = [&](auto add_tag) {
        constexpr bool ADD = decltype(add_tag)::value;
        // The 16 rows are software-pipelined by hand: the operand loads run EPI_DEPTH rows ahead of their use (ring in
        // registers, the twiddle ring is dead by now), otherwise every row waits for its own two loads with
        // vmcnt(0) -- which also drains the stores of the previous row -- and the epilogue costs 32 exposed round trips.
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
            u64 v0 = hp_harvey_lazy_nq(hp_sub_lazy(xv.x, x[2 * s], two_q), inv, invh, n0, n1);
            u64 v1 = hp_harvey_lazy_nq(hp_sub_lazy(xv.y, x[2 * s + 1], two_q), inv, invh, n0, n1);
            if (bgv) {
                v0 = hp_harvey_lazy_nq(v0, ql, qlh, n0, n1);
                v1 = hp_harvey_lazy_nq(v1, ql, qlh, n0, n1);
            }
            if (ADD) {
                v0 = hp_add_lazy(v0, av.x, two_q);
                v1 = hp_add_lazy(v1, av.y, two_q);
            }
            if (fin_on) {
                v0 = hp_harvey_lazy_nq(v0, fin, finh, n0, n1);
                v1 = hp_harvey_lazy_nq(v1, fin, finh, n0, n1);
            }
            d.store(voff + ((u32)s << 10), V2{v0, v1});
        }
        };
        if constexpr (FLAV == 2 || FLAV == 4) rows(std::true_type{});
        else if constexpr (FLAV == 1 || FLAV == 3) rows(std::false_type{});
        else if (has_add) rows(std::true_type{});
        else rows(std::false_type{});
    }
    TRACE_MARK();   // 9: stores issued
    TRACE_FLUSH();
}

template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW) k_ntt_fwd(HpNttJob job) {
    ntt_fwd_body<LOGN, false>(job, nullptr);
}

// forward NTT with the drop-last-prime prologue/epilogue fused in (HpDropArgs in kernel-argument memory)
template <int LOGN, int FLAV, bool SMALL>
__global__ void __launch_bounds__(Geo<LOGN>::T, Geo<LOGN>::MINW) k_ntt_fwd_drop(HpNttJob job, HpDropArgs da) {
    ntt_fwd_body<LOGN, true, FLAV, SMALL>(job, &da);
}

// ---- inverse kernel ----------------------------------------------------------------------------
// The middle pass of the inverse stages 31 x 32 twiddle pairs (15.5 KiB) in LDS whatever N is.  With one limb per workgroup
// that stage caps the occupancy of the small sizes (N = 4096: 16 + 15.5 KiB per two waves -> 5 workgroups = 2.5 waves per
// SIMD, VALUBusy 42 %).  So for N <= 8192 a workgroup transforms LPW limbs OF ONE MODULUS side by side and shares the stage:
// 512 threads, 4 N LPW + 15.5 KiB = 80 KiB of LDS, two workgroups = four waves per SIMD on a CU.
template <int LOGN> struct InvGeo {
    static constexpr int LPW = LOGN >= 14 ? 1 : (512 >> (LOGN - 5));   // limbs per workgroup: 2^11 -> 8, 2^12 -> 4, 2^13 -> 2
    static constexpr int TT = Geo<LOGN>::T * LPW;                      // threads per workgroup
    static constexpr bool STREAM_EPILOGUE = LOGN <= 13;                // see the end of k_ntt_inv
};

template <int LOGN, bool STRICT, bool PSCAL>
__global__ void __launch_bounds__(InvGeo<LOGN>::TT, Geo<LOGN>::MINW) k_ntt_inv(HpNttJob job) {
    using G = Geo<LOGN>;
    constexpr int LPW = InvGeo<LOGN>::LPW, TT = InvGeo<LOGN>::TT;
    __shared__ u32 lds_all[Addr<LOGN>::WORDS * LPW];
    __shared__ u64v2 lds_tw[31 * 32];
    TRACE_ENTRY
    const u32 sub = threadIdx.x / G::T, tid = threadIdx.x % G::T;   // limb of the workgroup, thread within the limb
    u32 *lds = lds_all + sub * Addr<LOGN>::WORDS;
    HpItem it;
    bool active = true;
    if (LPW == 1) {
        // (every inverse launch is HP_NTT_BATCH without groups: launch() rejects anything else)
        const u32 w = hp_xcd_remap(blockIdx.x, job.W);
        const u32 k = w / job.P, p = w % job.P;
        it.src = job.src + ((size_t)p * job.src_pstride + (size_t)k * job.src_kstride) * G::N;
        it.dst = job.dst + ((size_t)p * job.dst_pstride + k) * G::N;
        it.limb = k;
        it.poly = p;
    } else {
        // HP_NTT_BATCH only (every inverse launch is one): ceil(P / LPW) workgroups per modulus, modulus-major like the item
        // numbering; a group past the last polynomial re-reads the last one and stores nothing
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
    // the limb's constants and table pointers through the scalar cache (constant address space): as vector loads they would
    // queue behind the coefficient loads and the first pass could not start before nearly all of those are back
    const cptr_limb lp = (cptr_limb)(job.limbs + __builtin_amdgcn_readfirstlane(it.limb));
    const u64 q = lp->q, two_q = lp->two_q, nq = lp->neg_q;
    Addr<LOGN> ad;
    ad.init(tid);
    // stage the middle pass's twiddles (31 x 32 pairs; a few per thread when the workgroup has fewer than 992 threads)
    constexpr int NSTG = (31 * 32 + TT - 1) / TT;
    u64v2 stg[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const u32 e = threadIdx.x + (u32)i * TT;
        stg[i] = (e < 31u * 32u) ? ((gptr_u64x2)(lp->inv_k + 31))[e] : u64v2{0, 0};
    }

    TRACE_DECL
    TRACE_MARK();   // 0: entry (constants and staging loads issued)
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
#ifdef HP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TRACE_MARK();   // 1: coefficients have arrived
    exchange<LOGN, LAY_S, LAY_C, false>(x, lds, ad);
    TRACE_MARK();   // 2
    // pass A': levels 0..4 (pairs 1,2,4,8,16 apart), wave-uniform twiddles
    inv_pass<0, 4>(x, STab(lp->inv_k), 1u, 0u, nq, two_q);
    TRACE_MARK();   // 3
    exchange<LOGN, LAY_C, LAY_B, false>(x, lds, ad);
    TRACE_MARK();   // 4
    // The staged twiddles are read by other waves in pass B'; everything before it is wave-local, so this is the first point
    // where the waves have to meet.  (The barrier used to sit right after the loads: a workgroup's vector loads queue in wave
    // order behind each other at the CU's ~17 B/clk, the last wave gets its staging load out ~10 k cycles after the first, and
    // until then the first waves sat at the barrier with their coefficients long there: -5..8 % per launch at N = 32768.)
    __syncthreads();
    // pass B': levels 5..9, twiddles depend on j = tid & 31
    inv_pass<0, 4>(x, LTab(lds_tw), 32u, tid & 31u, nq, two_q);
    TRACE_MARK();   // 5
    exchange<LOGN, LAY_B, LAY_A, true>(x, lds, ad);
    TRACE_MARK();   // 6
    // pass C': levels 10..logN-1, per-thread twiddles
    // (twiddle ring of the last pass: 4 slots at N = 32768 / 8192; 3 at N = 16384 and 2 at N <= 4096, where the fourth costs spilled
    // registers -- 9 -> 4 and 19 -> 8 -- and buys nothing: +0.5 % / +3 % on the launch, less scratch traffic in WRITE_SIZE)
    constexpr int RING = LOGN == 14 ? 3 : LOGN <= 12 ? 2 : BTab::depth;
    inv_pass<G::PB, 4, BTab, RING>(x, BTab(lp->inv_k + 31 + 31 * 32), (u32)G::T, tid, nq, two_q);
    TRACE_MARK();   // 7
    if constexpr (InvGeo<LOGN>::STREAM_EPILOGUE) {
        // N <= 8192: in layout A a thread owns 2^PB >= 4 consecutive coefficients, so a 16-byte store instruction would write
        // a quarter or half of every cache line it touches: measured 1.9 x the algorithmic write traffic at N = 4096
        // (WRITE_SIZE, rocprofv3).  Fold in place, transpose once more (layout S, as the forward kernel stores), then the
        // psi^-i N^-1 pairs are read and the words written 16 contiguous bytes per lane, 1 KiB per wave instruction.
        const u32 k = lp->k, fix = lp->fix;
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = hp_shift_fold(x[r], q, k, fix);
        exchange<LOGN, LAY_A, LAY_S, true>(x, lds, ad);
        if (!active) return;   // (after the last barrier of the kernel)
        const size_t off = (((size_t)(tid >> 6)) << 11) + ((tid & 63u) << 1);
        const BTab sc(lp->inv_ref + G::N);   // pairs of coefficient off + (s << 7) + e: row = s, 128 pairs per row, lane part off + e
        u64 *d = it.dst + off;
        const u64 psc = job.post_scalar, psh = job.post_scalar_h;
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 2) {
            u64x2 f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = sc.at((u32)(s0 + (e >> 1)), 128u, (u32)off + (u32)(e & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 2 * s0 + e;
                u64 v = hp_harvey_lazy_nq(x[r], f[e].x, f[e].y, (u32)nq, (u32)(nq >> 32));
                if (PSCAL) v = hp_harvey_lazy_nq(v, psc, psh, (u32)nq, (u32)(nq >> 32));
                if (STRICT) v = hp_strict(v, q);
                x[r] = v;
            }
            st_stream(d + ((size_t)s0 << 7), V2{x[2 * s0], x[2 * s0 + 1]});
            st_stream(d + ((size_t)(s0 + 1) << 7), V2{x[2 * s0 + 2], x[2 * s0 + 3]});
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // fold, multiply by psi^-i * N^-1 (ntt.cpp:214-222), optional scalar + strict reduction, store (layout A)
    {
        const u32 k = lp->k, fix = lp->fix;
        const BTab sc(lp->inv_ref + G::N);   // pair of coefficient (kk << 10) + (tid << PB) + pp: row = kk, 1024 pairs per row
        u64 *d = it.dst + ((size_t)tid << G::PB);
        const u64 psc = job.post_scalar, psh = job.post_scalar_h;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 4) {
            u64x2 f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e, kk = r >> G::PB, pp = r & ((1 << G::PB) - 1);
                f[e] = sc.at((u32)kk, 1024u, (tid << G::PB) + (u32)pp);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                u64 v = hp_harvey_lazy_nq(hp_shift_fold(x[r], q, k, fix), f[e].x, f[e].y, (u32)nq, (u32)(nq >> 32));
                if (PSCAL) v = hp_harvey_lazy_nq(v, psc, psh, (u32)nq, (u32)(nq >> 32));
                if (STRICT) v = hp_strict(v, q);
                x[r] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        TRACE_MARK();   // 8: scaled
        if (!active) return;   // (after the last barrier of the kernel)
        if (G::PB == 0) {   // mirror of the forward load: lane pairs assemble 16-byte stores
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
    TRACE_MARK();   // 9: stores issued
    TRACE_FLUSH();
}

template <int LOGN> hipError_t launch(const HpNttJob &job, hipStream_t stream) {
    if (!job.inverse) {
        k_ntt_fwd<LOGN><<<job.W, Geo<LOGN>::T, 0, stream>>>(job);
        return hipGetLastError();
    }
    constexpr int LPW = InvGeo<LOGN>::LPW, TT = InvGeo<LOGN>::TT;
    if (job.mode != HP_NTT_BATCH || job.pair_moduli) return hipErrorNotSupported;
    const u32 grid = LPW == 1 ? job.W : job.L * ((job.P + LPW - 1) / LPW);
    if (job.use_post_scalar && job.strict) k_ntt_inv<LOGN, true, true><<<grid, TT, 0, stream>>>(job);
    else if (job.use_post_scalar) return hipErrorNotSupported;
    else if (job.strict) k_ntt_inv<LOGN, true, false><<<grid, TT, 0, stream>>>(job);
    else k_ntt_inv<LOGN, false, false><<<grid, TT, 0, stream>>>(job);
    return hipGetLastError();
}

} // namespace

#ifdef HP_TRACE
extern "C" int hp_debug_trace(u64 *out, size_t words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), words * sizeof(u64));
}
#endif

template <int LOGN>
static hipError_t launch_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream) {
    int flav = 0;
    if (!da.fin_on && !da.raw_input && !da.comb) {
        if (!da.addend || da.add_mask == 0) flav = 1;
        else if (da.add_mask == 3u) flav = 2;
        else if (da.add_mask == 1u && !da.dc.bgv) flav = 5;   // rotation / conjugation: += moved[0]
        if (flav && flav != 5 && da.dc.bgv) flav += 2;
    }
    const bool small = flav != 0 && da.small_rem != 0;   // (SMALL: the prologue's Barrett quotient is not needed, see HpDropArgs)
#define HP_DROP_LAUNCH(F, S) k_ntt_fwd_drop<LOGN, F, S><<<job.W, Geo<LOGN>::T, 0, stream>>>(job, da)
    if (flav == 1) { if (small) HP_DROP_LAUNCH(1, true); else HP_DROP_LAUNCH(1, false); }
    else if (flav == 2) { if (small) HP_DROP_LAUNCH(2, true); else HP_DROP_LAUNCH(2, false); }
    else if (flav == 3) { if (small) HP_DROP_LAUNCH(3, true); else HP_DROP_LAUNCH(3, false); }
    else if (flav == 4) { if (small) HP_DROP_LAUNCH(4, true); else HP_DROP_LAUNCH(4, false); }
    else if (flav == 5) { if (small) HP_DROP_LAUNCH(5, true); else HP_DROP_LAUNCH(5, false); }
    else HP_DROP_LAUNCH(0, false);
#undef HP_DROP_LAUNCH
    return hipGetLastError();
}

hipError_t hp_launch_ntt_fast_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    switch (job.logn) {
    case 11: return launch_drop<11>(job, da, stream);
    case 12: return launch_drop<12>(job, da, stream);
    case 13: return launch_drop<13>(job, da, stream);
    case 14: return launch_drop<14>(job, da, stream);
    case 15: return launch_drop<15>(job, da, stream);
    default: return hipErrorNotSupported;
    }
}

hipError_t hp_launch_ntt_fast(const HpNttJob &job, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    switch (job.logn) {
    case 11: return launch<11>(job, stream);
    case 12: return launch<12>(job, stream);
    case 13: return launch<13>(job, stream);
    case 14: return launch<14>(job, stream);
    case 15: return launch<15>(job, stream);
    default: return hipErrorNotSupported;
    }
}
