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
// a third of the tiled kernels', latency a quarter: hp_ctx.cpp picks this path when a launch has at most HP_SPLIT_MAX_ITEMS limbs.
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

// LDS-only barrier: the twiddle loads of the NEXT stage stay in flight across it (__syncthreads would wait for them)
HP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Bfly {
    u32 l, h, tw;   // LDS slots of the pair, index of its twiddle in the limb's reference-order table
};

// butterfly number b (0 .. 1023) of step `step` of the kernel's run of stages
template <bool INVERSE, bool COLS> HP_DEV Bfly bfly_of(u32 logn, const TileMap<COLS> &tm, u32 step, u32 b) {
    Bfly f;
    if (!INVERSE && COLS) {
        // stages s = 1 .. logn - 8: row gap 2^lg, twiddle (1 << (s - 1)) + (row block)                  ntt.cpp:155-169
        const u32 s = step + 1, lg = (logn - 8) - s;
        const u32 col = b & (tm.ccount - 1), pr = b >> tm.lc;
        const u32 blk = pr >> lg, j = pr & ((1u << lg) - 1);
        const u32 r0 = (blk << (lg + 1)) | j;
        f.l = (r0 << tm.lc) + col; f.h = f.l + (tm.ccount << lg);
        f.tw = (1u << (s - 1)) + blk;
    } else if (!INVERSE) {
        // stages s = logn - 7 .. logn: gaps 128 .. 1 inside the tile; the block number is global
        const u32 s = logn - 7 + step, lg = logn - s, gap = 1u << lg;
        const u32 blk = b >> lg, j = b & (gap - 1);
        f.l = (blk << (lg + 1)) | j; f.h = f.l + gap;
        f.tw = (1u << (s - 1)) + ((tm.base + f.l) >> (lg + 1));
    } else if (!COLS) {
        // gaps 1 .. 128 (s = 0 .. 7): twiddle level s, entry bitrev_s(i mod 2^s)                           ntt.cpp:178-213
        const u32 sft = step, gap = 1u << sft;
        const u32 blk = b >> sft, c = b & (gap - 1);
        f.l = (blk << (sft + 1)) | c; f.h = f.l + gap;
        f.tw = (gap - 1) + (sft ? (__brev(c) >> (32 - sft)) : 0u);
    } else {
        // gaps 256 .. N/2 (s = 8 .. logn - 1): row gap 2^(s-8); the low bits of the global index select the twiddle
        const u32 sft = 8 + step, lg = step;
        const u32 col = b & (tm.ccount - 1), pr = b >> tm.lc;
        const u32 blk = pr >> lg, j = pr & ((1u << lg) - 1);
        const u32 r0 = (blk << (lg + 1)) | j;
        f.l = (r0 << tm.lc) + col; f.h = f.l + (tm.ccount << lg);
        const u32 c = (j << 8) | (tm.base + col);          // (global index of the low partner) mod 2^s
        f.tw = ((1u << sft) - 1) + (__brev(c) >> (32 - sft));
    }
    return f;
}

// DROP (forward only): the drop-last-prime step around the transform, as the tiled k_ntt_fwd_drop has it -- the first launch reads
// the strict last-limb coefficients and applies Barrett + centring [+ * t] while loading (rescaling.cpp:54-69, mod_switch.cpp:52-69:
// k_drop_rem's arithmetic), the second finishes with out = ((x - NTT(rem)) * inv) [* (q_last mod t)] [+ addend] while storing
// (rescaling.cpp:72-74, mod_switch.cpp:72-76, ckks/arith.cpp:70-71: k_drop_fin's arithmetic); the rows between the launches are scratch
template <bool INVERSE, bool COLS, bool FIRST, bool DROP>
HP_DEV void ntt_split_body(const HpNttJob &job, const HpDropArgs *da) {
    __shared__ __attribute__((aligned(16))) u64 buf[SPLIT_TILE];
    constexpr int PER = SPLIT_TILE / SPLIT_THREADS, BPT = PER / 2;   // coefficients and butterflies per thread and stage
    const u32 logn = job.logn, n = 1u << logn, tiles = n / SPLIT_TILE;
    const u32 w = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    HpItem it;
    if (!hp_decode_item(job, w, it)) return;
    const HpLimb m = job.limbs[it.limb];
    const gptr_tw table = (gptr_tw)(INVERSE ? m.inv_ref : m.fwd_ref);
    const TileMap<COLS> tm(logn, tile);
    const u64 *in = FIRST ? it.src : it.dst;   // the second launch works on what the first one left in the destination row
    const u32 steps = COLS ? logn - 8 : 8;
    // everything a thread needs from memory before its first butterfly is issued at once: its coefficients, the twiddles of the
    // first stage and (inverse, last launch) the psi^-i N^-1 pairs of its outputs
    u64 x[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) x[e] = in[tm.global(threadIdx.x + e * SPLIT_THREADS)];
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
    Bfly cur[BPT];
    u64x2 tw[BPT];
#pragma unroll
    for (int t = 0; t < BPT; ++t) {
        cur[t] = bfly_of<INVERSE, COLS>(logn, tm, 0, threadIdx.x + t * SPLIT_THREADS);
        tw[t] = ld_tw(table, cur[t].tw);
    }
    u64x2 sc[PER];
    if (INVERSE && COLS) {
#pragma unroll
        for (int e = 0; e < PER; ++e) sc[e] = ld_tw(table, n + tm.global(threadIdx.x + e * SPLIT_THREADS));   // ntt.cpp:214-222
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) buf[threadIdx.x + e * SPLIT_THREADS] = x[e];
    lds_barrier();
    for (u32 step = 0; step < steps; ++step) {
        // the next stage's twiddles go out before this stage's arithmetic: their L2 latency hides behind it and the barrier
        Bfly nxt[BPT];
        u64x2 ntw[BPT];
        const bool more = step + 1 < steps;
#pragma unroll
        for (int t = 0; t < BPT; ++t) {
            nxt[t] = bfly_of<INVERSE, COLS>(logn, tm, more ? step + 1 : step, threadIdx.x + t * SPLIT_THREADS);
            ntw[t] = ld_tw(table, nxt[t].tw);
        }
#pragma unroll
        for (int t = 0; t < BPT; ++t) {
            u64 lo = buf[cur[t].l], hi = buf[cur[t].h];
            hp_butterfly(lo, hi, tw[t].x, tw[t].y, m.q, m.two_q);
            buf[cur[t].l] = lo; buf[cur[t].h] = hi;
        }
        lds_barrier();
#pragma unroll
        for (int t = 0; t < BPT; ++t) { cur[t] = nxt[t]; tw[t] = ntw[t]; }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const u32 i = threadIdx.x + e * SPLIT_THREADS, g = tm.global(i);
        u64 v = buf[i];
        if (!INVERSE && !COLS) v = hp_shift_fold(v, m.q, m.k, m.fix);                         // ntt.cpp:171-175 after the last stage
        if (DROP && !FIRST) {
            const u32 k = it.limb, p2 = it.poly;
            const u64 *xs = da->x + ((size_t)p2 * da->L + k) * n;
            v = hp_sub_lazy(xs[g], v, m.two_q);
            v = hp_harvey_lazy(v, da->dc.inv[k], da->dc.inv_h[k], m.q);
            if (da->dc.bgv) v = hp_harvey_lazy(v, da->dc.qlt[k], da->dc.qlt_h[k], m.q);
            if (da->addend && ((da->add_mask >> (p2 & 1)) & 1u))
                v = hp_add_lazy(v, da->addend[((size_t)(p2 >> 1) * da->add_ct_stride + (size_t)(p2 & 1) * da->add_poly_stride + k) * n + g], m.two_q);
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
