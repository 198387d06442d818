// hp_ntt_split.hip -- a limb transform SPLIT over several workgroups, for launches with fewer limbs than the GPU has CUs.
//
// The tiled kernels (hp_ntt_fast.hip) give one limb to one workgroup = one CU: 4 waves per SIMD x ~4 500 - 5 000 VALU instructions =
// 36 - 44 us per limb whatever else the GPU does.  A batch of 256 ciphertexts hides that behind 25 600 limbs per launch; ONE
// ciphertext through hehub's one-call-per-ciphertext interface (ckks.h:270-313) does not: its launches carry 2 .. 100 limbs, and a
// C3 hom-mult is a chain of twelve of them = 0.33 ms with nine tenths of the GPU idle.  Here a limb of N = 2^logn coefficients is cut
// into N / 2048 tiles and transformed by TWO launches of 256-thread workgroups, each running a run of the reference's stages on
// one tile in LDS (16 KiB):
//
//   forward (ntt.cpp:155-176)   launch 1: stages 1 .. logn-8 (gaps N/2 .. 256) on COLUMN tiles (all rows of 2048 / (N/256) columns:
//                                         a butterfly pairs two rows of one column, its twiddle depends on the row block only);
//                               launch 2: stages logn-7 .. logn (gaps 128 .. 1) on CONTIGUOUS tiles of 2048, then the fold
//   inverse (ntt.cpp:178-223)   launch 1: stages of gaps 1 .. 128 on contiguous tiles; launch 2: gaps 256 .. N/2 on column tiles,
//                                         then fold, psi^-i N^-1, the optional scalar (mod_switch.cpp:49), reduce_strict
//
// Every butterfly is the reference's, with the reference's twiddle (hp_ntt_generic.hip's indexing): the words are identical to the
// tiled kernels' and to hehub's.  The intermediate limb crosses L2 once (the launches are small: it never reaches HBM).  Throughput is
// a third of the tiled kernels', latency a fifth: hp_ctx.cpp picks this path when a launch has at most HP_SPLIT_MAX_ITEMS limbs.
// Inside a launch the stages run in ROUNDS of up to three in registers (see below): 100 limbs of N = 32768 forward 18 + 16 us
// (24 + 22 with one stage per LDS exchange), 10 limbs inverse 5 + 7.5 us (8.5 + 11): batch-1 C3 hom-mult 0.162 -> 0.124 ms.
#include "hp_kernels.h"
#include "hp_ntt_job.h"

namespace {

typedef u64 __attribute__((ext_vector_type(2))) u64v2s;
typedef const u64v2s __attribute__((address_space(1))) * gptr_tw;
HP_DEV u64x2 ld_tw(gptr_tw p, u32 i) {   // one global_load_dwordx4 (a generic pointer would make it a flat load: vmcnt AND lgkmcnt)
    const u64v2s v = p[i];
    return u64x2{v.x, v.y};
}

constexpr int SPLIT_THREADS = 256;
constexpr int SPLIT_TILE = 2048;   // coefficients per workgroup

// COLS: the tile is all rows of a set of columns of the limb seen as [N / 256][256]; otherwise 2048 contiguous coefficients
template <bool COLS> struct TileMap {
    u32 lc, ccount, base;   // COLS: ccount = 2^lc columns (all N / 256 rows of them) from column `base`; else base = first coefficient
    HP_DEV TileMap(u32 logn, u32 tile) {
        if (COLS) {
            lc = 19 - logn;            // 2048 / (N / 256) columns
            ccount = 1u << lc;
            base = tile << lc;
        } else {
            lc = 0; ccount = 0;
            base = tile * SPLIT_TILE;
        }
    }
    // global coefficient index of local slot i (COLS: slot = row * ccount + col)
    HP_DEV u32 global(u32 i) const { return COLS ? ((i >> lc) << 8) + base + (i & (ccount - 1)) : base + i; }
};

// LDS-only barrier: the twiddle loads of the NEXT round stay in flight across it (__syncthreads would wait for them)
HP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A launch runs its stages in ROUNDS of up to three: a round covers K consecutive bits [p, p + K) of the 11-bit slot index, a thread
// holds the 2^K slots that differ in those bits (8 / 2^K such groups) and runs the K stages on them in registers -- the hand-scheduled
// dual butterfly of the tiled kernels, 16 instructions each, no index arithmetic per butterfly -- and only then goes through LDS
// (one exchange per round instead of one per stage; the first version of this file, a stage per exchange, spent 40 VALU instructions
// per butterfly and a third of its LDS cycles on bank conflicts: profiles/r05n_split_pmc_summary.txt).
// Slot of register e (r = low K bits: the round's bits, g = the rest: which group) of thread tid:
template <int K> HP_DEV u32 slot_of(u32 tid, u32 p, int e) {
    const u32 r = (u32)e & ((1u << K) - 1u), u = tid | (((u32)e >> K) << 8);
    return (u & ((1u << p) - 1u)) | (r << p) | ((u >> p) << (p + K));
}
HP_DEV u32 swz(u32 slot) { return slot ^ ((slot >> 4) & 15u); }   // LDS position of a slot (strided rounds would hit few banks)

// index, in the limb's reference-order table, of the twiddle of the butterfly whose LOW partner sits in slot sl and whose partners
// differ in slot bit b (hp_ntt_generic.hip's indexing, as the launches see it):
//   forward, column tiles       stage s = 11 - b, row gap 2^(b - lc):   (1 << (s - 1)) + (row block)                 ntt.cpp:155-169
//   forward, contiguous tiles   stage s = logn - b, gap 2^b:            (1 << (s - 1)) + (global block)
//   inverse, contiguous tiles   gap 2^b:   level b, entry bitrev_b(i mod 2^b)                                       ntt.cpp:178-213
//   inverse, column tiles       gap 2^(8 + b - lc): level 8 + b - lc, entry bitrev(global index of the low partner mod the gap)
template <bool INVERSE, bool COLS> HP_DEV u32 tw_index(u32 logn, const TileMap<COLS> &tm, u32 b, u32 sl) {
    if (!INVERSE && COLS) return (1u << (10u - b)) + (sl >> (b + 1));
    if (!INVERSE) return (1u << (logn - b - 1)) + ((tm.base + sl) >> (b + 1));
    if (!COLS) return ((1u << b) - 1u) + (b ? (__brev(sl & ((1u << b) - 1u)) >> (32 - b)) : 0u);
    const u32 step = b - tm.lc, sft = 8 + step;
    const u32 c = (((sl >> tm.lc) & ((1u << step) - 1u)) << 8) | (tm.base + (sl & (tm.ccount - 1u)));
    return ((1u << sft) - 1u) + (__brev(c) >> (32 - sft));
}

// stage j of a round pairs the registers that differ in bit rb of their index (forward: the round's bits from the top, inverse:
// from the bottom); its twiddles: one per value of the round's bits above (forward) / below (inverse) rb, per group
template <bool INVERSE, int K> constexpr int stage_bit(int j) { return INVERSE ? j : K - 1 - j; }
template <bool INVERSE, int K> constexpr int tw_slot(int e_lo, int j) {
    const int rb = stage_bit<INVERSE, K>(j), r = e_lo & ((1 << K) - 1), g = e_lo >> K;
    return g * ((1 << K) - 1) + ((1 << j) - 1) + (INVERSE ? (r & ((1 << rb) - 1)) : (r >> (rb + 1)));
}

template <bool INVERSE, bool COLS, int K>
HP_DEV void round_twiddles(u64x2 (&tw)[7], gptr_tw table, u32 logn, const TileMap<COLS> &tm, u32 p) {
#pragma unroll
    for (int g = 0; g < (8 >> K); ++g)
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int t = 0; t < (1 << j); ++t) {
                const int rb = stage_bit<INVERSE, K>(j);
                const int e_lo = (g << K) | (INVERSE ? t : (t << (rb + 1)));   // a representative: bit rb clear, the other bits free
                tw[g * ((1 << K) - 1) + ((1 << j) - 1) + t] = ld_tw(table, tw_index<INVERSE, COLS>(logn, tm, p + rb, slot_of<K>(threadIdx.x, p, e_lo)));
            }
}

template <bool INVERSE, int K>
HP_DEV void round_stages(u64 (&x)[8], const u64x2 (&tw)[7], u64 two_q, u32 n0, u32 n1) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int rb = stage_bit<INVERSE, K>(j);
#pragma unroll
        for (int d = 0; d < 4; d += 2) {   // the stage's four pairs, two at a time (pair d: the 2-bit number d with a zero inserted at bit rb)
            const int a = ((d >> rb) << (rb + 1)) | (d & ((1 << rb) - 1)), b = (((d + 1) >> rb) << (rb + 1)) | ((d + 1) & ((1 << rb) - 1));
            const u64x2 wa = tw[tw_slot<INVERSE, K>(a, j)], wb = tw[tw_slot<INVERSE, K>(b, j)];
            hp_butterfly2_nq(x[a], x[a | (1 << rb)], x[b], x[b | (1 << rb)], wa.x, wa.y, wb.x, wb.y, two_q, n0, n1);
        }
    }
}

// DROP (forward only): the drop-last-prime step around the transform, as the tiled k_ntt_fwd_drop has it -- the first launch reads
// the strict last-limb coefficients and applies Barrett + centring [+ * t] while loading (rescaling.cpp:54-69, mod_switch.cpp:52-69:
// k_drop_rem's arithmetic), the second finishes with out = ((x - NTT(rem)) * inv) [* (q_last mod t)] [+ addend] while storing
// (rescaling.cpp:72-74, mod_switch.cpp:72-76, ckks/arith.cpp:70-71: k_drop_fin's arithmetic); the rows between the launches are scratch
//
// Rounds KA @ pA, KB @ pB, KC @ pC in the order they run (KC = 0: two rounds).  Coefficients are loaded straight into the first
// round's registers when its slots lie in runs of >= 16 in memory (pA >= 4), otherwise in the linear order (slot = tid + 256 e) and
// through LDS; the same at the end.
template <bool INVERSE, bool COLS, bool FIRST, bool DROP, int KA, int KB, int KC>
HP_DEV void ntt_split_rounds(const HpNttJob &job, const HpDropArgs *da, u64 *buf, u32 pA, u32 pB, u32 pC) {
    constexpr int PER = SPLIT_TILE / SPLIT_THREADS;
    static_assert(PER == 8, "eight coefficients per thread");
    constexpr int KL = KC ? KC : KB;     // the last round
    const u32 pL = KC ? pC : pB;
    const u32 logn = job.logn, n = 1u << logn;
    const u32 w = blockIdx.x >> (logn - 11), tile = blockIdx.x & ((n / SPLIT_TILE) - 1u), tid = threadIdx.x;   // N / 2048 tiles per limb
    HpItem it;
    if (!hp_decode_item(job, w, it)) return;
    const HpLimb m = job.limbs[it.limb];
    const u32 n0 = (u32)m.neg_q, n1 = (u32)(m.neg_q >> 32);
    const gptr_tw table = (gptr_tw)(INVERSE ? m.inv_ref : m.fwd_ref);
    const TileMap<COLS> tm(logn, tile);
    const u64 *in = FIRST ? it.src : it.dst;   // the second launch works on what the first one left in the destination row
    const bool lin_in = pA < 4, lin_out = pL < 4;
    // everything a thread needs from memory before its first butterfly is issued at once: its coefficients, the twiddles of the
    // first round and (inverse, last launch) the psi^-i N^-1 pairs of its outputs
    u64 x[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) x[e] = in[tm.global(lin_in ? tid + (u32)e * SPLIT_THREADS : slot_of<KA>(tid, pA, e))];
    u64x2 tw[7];
    round_twiddles<INVERSE, COLS, KA>(tw, table, logn, tm, pA);
    u64x2 sc[PER];
    if (INVERSE && COLS) {
#pragma unroll
        for (int e = 0; e < PER; ++e)
            sc[e] = ld_tw(table, n + tm.global(lin_out ? tid + (u32)e * SPLIT_THREADS : slot_of<KL>(tid, pL, e)));   // ntt.cpp:214-222
    }
    // (drop, second launch) what the epilogue combines the transform with -- the limb's own coefficients and the optional addend -- is
    // asked for now as well: loaded in the store loop, their latency sat exposed at the end of a launch that is all latency
    u64 xs_pre[PER], add_pre[PER];
    bool with_addend = false;
    if (DROP && !FIRST) {
        const u32 k = it.limb, p2 = it.poly;
        const u64 *xs = da->x + ((size_t)p2 * da->L + k) * n;
        with_addend = da->addend && ((da->add_mask >> (p2 & 1)) & 1u);
        const u64 *ad = with_addend ? da->addend + ((size_t)(p2 >> 1) * da->add_ct_stride + (size_t)(p2 & 1) * da->add_poly_stride + k) * n : xs;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const u32 g = tm.global(lin_out ? tid + (u32)e * SPLIT_THREADS : slot_of<KL>(tid, pL, e));
            xs_pre[e] = xs[g];
            add_pre[e] = ad[g];
        }
    }
    if (DROP && FIRST) {
        const u32 k = it.limb;
        const u64 bump = m.q - da->dc.r[k];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const u64 c = x[e];
            u64 v = hp_strict(hp_barrett_lazy(c, m.q, m.barrett_c), m.q);
            if (c >= da->dc.half_q_last) v += bump;
            if (da->dc.bgv) v = hp_harvey_lazy(v, da->dc.t[k], da->dc.t_h[k], m.q);
            x[e] = v;
        }
    }
    if (lin_in) {
#pragma unroll
        for (int e = 0; e < PER; ++e) buf[swz(tid + (u32)e * SPLIT_THREADS)] = x[e];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < PER; ++e) x[e] = buf[swz(slot_of<KA>(tid, pA, e))];
    }
    round_stages<INVERSE, KA>(x, tw, m.two_q, n0, n1);
    // the next round's twiddles go out before the exchange: their L2 latency hides behind it and the barrier.  A thread writes the
    // slots it read (nobody else's), so one barrier per exchange is enough.
    round_twiddles<INVERSE, COLS, KB>(tw, table, logn, tm, pB);
#pragma unroll
    for (int e = 0; e < PER; ++e) buf[swz(slot_of<KA>(tid, pA, e))] = x[e];
    lds_barrier();
#pragma unroll
    for (int e = 0; e < PER; ++e) x[e] = buf[swz(slot_of<KB>(tid, pB, e))];
    round_stages<INVERSE, KB>(x, tw, m.two_q, n0, n1);
    if constexpr (KC != 0) {
        round_twiddles<INVERSE, COLS, KC>(tw, table, logn, tm, pC);
#pragma unroll
        for (int e = 0; e < PER; ++e) buf[swz(slot_of<KB>(tid, pB, e))] = x[e];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < PER; ++e) x[e] = buf[swz(slot_of<KC>(tid, pC, e))];
        round_stages<INVERSE, KC>(x, tw, m.two_q, n0, n1);
    }
    if (lin_out) {
#pragma unroll
        for (int e = 0; e < PER; ++e) buf[swz(slot_of<KL>(tid, pL, e))] = x[e];
        lds_barrier();
#pragma unroll
        for (int e = 0; e < PER; ++e) x[e] = buf[swz(tid + (u32)e * SPLIT_THREADS)];
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const u32 g = tm.global(lin_out ? tid + (u32)e * SPLIT_THREADS : slot_of<KL>(tid, pL, e));
        u64 v = x[e];
        if (!INVERSE && !COLS) v = hp_shift_fold(v, m.q, m.k, m.fix);                         // ntt.cpp:171-175 after the last stage
        if (DROP && !FIRST) {
            const u32 k = it.limb, p2 = it.poly;
            v = hp_sub_lazy(xs_pre[e], v, m.two_q);
            v = hp_harvey_lazy(v, da->dc.inv[k], da->dc.inv_h[k], m.q);
            if (da->dc.bgv) v = hp_harvey_lazy(v, da->dc.qlt[k], da->dc.qlt_h[k], m.q);
            if (with_addend) v = hp_add_lazy(v, add_pre[e], m.two_q);
            da->out[((size_t)p2 * da->out_stride + k) * n + g] = v;
            continue;
        }
        if (INVERSE && COLS) {
            v = hp_harvey_lazy(hp_shift_fold(v, m.q, m.k, m.fix), sc[e].x, sc[e].y, m.q);     // ntt.cpp:214-222
            if (job.use_post_scalar) v = hp_harvey_lazy(v, job.post_scalar, job.post_scalar_h, m.q);
            if (job.strict) v = hp_strict(v, m.q);
        }
        it.dst[g] = v;
    }
}

// The rounds of each launch.  Contiguous tiles: the eight stages of gaps 128 .. 1 = slot bits 7 .. 0 (forward: 3 + 3 + 2 from the
// top; inverse: 3 + 3 + 2 from the bottom).  Column tiles: the logn - 8 stages of slot bits 10 .. lc (lc = 19 - logn): the round
// of bits 8 .. 10 -- whose slots are the linear order -- first (forward) / last (inverse), the remaining logn - 11 .. bits as one
// or two more rounds.
template <bool INVERSE, bool COLS, bool FIRST, bool DROP>
HP_DEV void ntt_split_body(const HpNttJob &job, const HpDropArgs *da) {
    __shared__ __attribute__((aligned(16))) u64 buf[SPLIT_TILE];   // (one buffer: every case of the switch below would otherwise bring its own)
    if constexpr (!COLS) {
        if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 2>(job, da, buf, 0, 3, 6);
        else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 2>(job, da, buf, 5, 2, 0);
    } else {
        switch (job.logn) {   // (uniform: one case per launch)
        case 12: if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 1, 3, 0>(job, da, buf, 7, 8, 0); else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 1, 0>(job, da, buf, 8, 7, 0); break;
        case 13: if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 2, 3, 0>(job, da, buf, 6, 8, 0); else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 2, 0>(job, da, buf, 8, 6, 0); break;
        case 14: if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 0>(job, da, buf, 5, 8, 0); else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 0>(job, da, buf, 8, 5, 0); break;
        case 15: if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 1, 3, 3>(job, da, buf, 4, 5, 8); else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 1>(job, da, buf, 8, 5, 4); break;
        default: if (INVERSE) ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 2, 3, 3>(job, da, buf, 3, 5, 8); else ntt_split_rounds<INVERSE, COLS, FIRST, DROP, 3, 3, 2>(job, da, buf, 8, 5, 3); break;
        }
    }
}

template <bool INVERSE, bool COLS, bool FIRST>
__global__ void __launch_bounds__(SPLIT_THREADS) k_ntt_split(HpNttJob job) {
    ntt_split_body<INVERSE, COLS, FIRST, false>(job, nullptr);
}
template <bool COLS, bool FIRST>
__global__ void __launch_bounds__(SPLIT_THREADS) k_ntt_split_drop(HpNttJob job, HpDropArgs da) {
    ntt_split_body<false, COLS, FIRST, true>(job, &da);
}

} // namespace

// the fused drop of a few limbs: job = HP_NTT_BATCH over the remaining limbs with src = clast [P2][n] (src_pstride 1, src_kstride 0)
// and dst = scratch rows [P2][kc][n]; plain drops only (no raw input, no final multiplication, no combination: those are the hybrid
// key switch's, which keeps the tiled kernel)
hipError_t hp_launch_ntt_split_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    if (job.logn < 12 || job.logn > 16 || job.limbs_a || job.inverse || job.mode != HP_NTT_BATCH || job.pair_moduli || !job.dst ||
        da.raw_input || da.fin_on || da.comb)
        return hipErrorNotSupported;
    const dim3 grid(job.W * ((1u << job.logn) / SPLIT_TILE));
    k_ntt_split_drop<true, true><<<grid, SPLIT_THREADS, 0, stream>>>(job, da);
    k_ntt_split_drop<false, false><<<grid, SPLIT_THREADS, 0, stream>>>(job, da);
    return hipGetLastError();
}

// logn in [12, 16] (at least two tiles per limb and eight block stages); plain u64 rows only (no packed digit rows, no level A)
hipError_t hp_launch_ntt_split(const HpNttJob &job, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    if (job.logn < 12 || job.logn > 16 || job.limbs_a || job.pack_mask || job.pack40_mask) return hipErrorNotSupported;
    const u32 tiles = (1u << job.logn) / SPLIT_TILE;
    const dim3 grid(job.W * tiles);
    if (!job.inverse) {
        k_ntt_split<false, true, true><<<grid, SPLIT_THREADS, 0, stream>>>(job);
        k_ntt_split<false, false, false><<<grid, SPLIT_THREADS, 0, stream>>>(job);
    } else {
        k_ntt_split<true, false, true><<<grid, SPLIT_THREADS, 0, stream>>>(job);
        k_ntt_split<true, true, false><<<grid, SPLIT_THREADS, 0, stream>>>(job);
    }
    return hipGetLastError();
}
