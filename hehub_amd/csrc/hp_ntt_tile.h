// hp_ntt_tile.h -- the register / LDS tiling shared by the tiled transforms (hp_ntt_fast.hip: exact Harvey butterflies, parity
// level B; hp_ntt_a.hip: error-free FP64 residue butterflies, parity level A): geometry, twiddle-table readers, LDS exchange
// layouts, streaming HBM accesses, the forward kernel's load order.  See the head of hp_ntt_fast.hip for the schedule.
#pragma once
#include "hp_kernels.h"
#include "hp_ntt_job.h"
#include <type_traits>

namespace {


template <int LOGN> struct Geo {
    static constexpr int A = LOGN - 10;          // stages of pass A (1..5)
    static constexpr int PB = 5 - A;             // passenger bits of pass A's register index
    static constexpr int T = 1 << (LOGN - 5);    // threads per workgroup
    static constexpr int N = 1 << LOGN;
    static constexpr int MINW = (T >= 1024) ? 4 : 4;   // waves per SIMD wanted (<= 128 VGPRs)
};

HP_DEV u32 lo32(u64 v) { return (u32)v; }
HP_DEV u32 hi32(u64 v) { return (u32)(v >> 32); }
HP_DEV u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// ---- passes ---------------------------------------------------------------------------------
// A pass runs up to five radix-2 stages on the 5-bit register index.  Its twiddles are numbered by
// "slot" 0..30 in the order they are consumed:
//   forward: stage on register bit b = 4 - floor(log2(slot+1)); the pairs (r, r | 1<<b) that use the
//            slot are those with r >> (b+1) == slot + 1 - 2^(4-b)
//   inverse: stage on register bit b = floor(log2(slot+1));      pairs with (r & (2^b - 1)) == slot + 1 - 2^b
// Twiddle loads run TW_DEPTH slots ahead of their use through a small register ring; scheduling
// barriers keep the compiler from hoisting all 31 loads (124 VGPRs) to the top of the pass, which
// would push the kernel past the 128-VGPR budget of 4 waves per SIMD.
constexpr int TW_DEPTH = 4;     // deeper rings spill (measured)

typedef u64 __attribute__((ext_vector_type(2))) u64v2;
typedef const u64v2 __attribute__((address_space(1))) * gptr_u64x2;
typedef const u64v2 __attribute__((address_space(3))) * lptr_u64x2;
HP_DEV u64x2 gload(gptr_u64x2 p, size_t i) {   // one global_load_dwordx4
    const u64v2 v = p[i];
    return u64x2{v.x, v.y};
}

// Where a pass finds its (w, w') pairs.
//  GTab: a per-modulus table in global memory.  The table pointer comes out of a struct in memory, so the
//        compiler only knows it as a generic (flat) pointer; flat loads bump both vmcnt and lgkmcnt and force
//        full s_waitcnt 0 waits, which would serialise the prefetch ring behind L2 latency -- so it is cast to
//        the global address space and the loads become counted global_load_dwordx4.
//  LTab: a copy of the pass table staged in LDS by the workgroup (the per-1024-block tables of the middle
//        pass: 31 * 2^A pairs forward, 31 * 32 pairs inverse).  Each entry is needed by exactly one half-wave,
//        so from global memory every read would be an L1 miss; from LDS it is a broadcast ds_read_b128.
struct GTab {
    static constexpr bool scalar = false;
    static constexpr int depth = TW_DEPTH;   // slots of L2 latency to cover
    gptr_u64x2 p;
    HP_DEV explicit GTab(const u64x2 *generic) : p((gptr_u64x2)generic) {}
    HP_DEV u64x2 operator()(u32 i) const { const u64v2 v = p[i]; return u64x2{v.x, v.y}; }
    HP_DEV u64x2 at(u32 row, u32 ncls, u32 cls) const { return (*this)(row * ncls + cls); }
};
// BTab: the same table read with BUFFER loads: address = descriptor base (SGPRs) + per-launch scalar offset (SGPR, the slot) + one
//       32-bit lane offset (VGPR, computed once) -- no 64-bit address arithmetic on the vector ALU per slot (the global form
//       costs a v_add_co / v_addc pair per load: 66 instructions per thread in the last pass).
typedef u32 __attribute__((ext_vector_type(4))) v4u32;
struct BTab {
    static constexpr bool scalar = false;
    static constexpr int depth = TW_DEPTH;
    __amdgpu_buffer_rsrc_t rsrc;
    // raw buffer over the whole address range above the table (no bounds clamp wanted), gfx9-family data format word
    HP_DEV explicit BTab(const u64x2 *uniform_base)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc((void *)uniform_base, 0, 0x7fffffff, 0x00020000)) {}
    // entry (row * ncls + cls) of 16-byte pairs: row * ncls is wave-uniform, cls per lane
    HP_DEV u64x2 at(u32 row, u32 ncls, u32 cls) const {
        const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, cls << 4, (row * ncls) << 4, 0);
        return u64x2{((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
    }
};
// STab: wave-uniform entries of a global table (the first pass of a transform): constant address space, so the
//       loads are scalar (s_load_dwordx4 through the scalar cache) and take no vector-memory slots.
typedef const u64v2 __attribute__((address_space(4))) * cptr_u64x2;
typedef const HpLimb __attribute__((address_space(4))) * cptr_limb;
constexpr int STAB_DEPTH = 4;   // 2 / 8 / 16 measured the same or worse
struct STab {
    static constexpr bool scalar = true;
    static constexpr int depth = STAB_DEPTH;   // held in SGPRs; SMEM returns out of order, so every use waits for all of them
    cptr_u64x2 p;
    HP_DEV explicit STab(const u64x2 *generic) : p((cptr_u64x2)generic) {}
    HP_DEV u64x2 operator()(u32 i) const { const u64v2 v = p[i]; return u64x2{v.x, v.y}; }
    HP_DEV u64x2 at(u32 row, u32 ncls, u32 cls) const { return (*this)(row * ncls + cls); }
};
struct LTab {
    static constexpr bool scalar = false;
    static constexpr int depth = 2;          // LDS latency is short
    lptr_u64x2 p;
    HP_DEV explicit LTab(const u64v2 *shared) : p((lptr_u64x2)shared) {}
    HP_DEV u64x2 operator()(u32 i) const { const u64v2 v = p[i]; return u64x2{v.x, v.y}; }
    HP_DEV u64x2 at(u32 row, u32 ncls, u32 cls) const { return (*this)(row * ncls + cls); }
};

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// BTab / StreamBuf address with 32-bit byte offsets into descriptors without a bounds clamp: the largest offsets are
// (31 rows x 1024 classes + 1023) pairs of 16 bytes for a table and one limb (8 N bytes) for a coefficient row
static_assert((31ull * 1024 + 1024) * 16 < (1ull << 31) && (8ull << 15) < (1ull << 31), "buffer offsets must stay below 2^31");

// register index of the o-th butterfly of slot s (compile-time): forward idx = bits above b, o = bits below b;
// inverse the other way round
template <bool FWD> constexpr int slot_reg(int s, int o) {
    const int lg = ilog2c(s + 1), b = FWD ? 4 - lg : lg, idx = s + 1 - (1 << lg);
    return FWD ? ((idx << (b + 1)) | o) : ((o << (b + 1)) | idx);
}
template <bool FWD> constexpr int slot_bit(int s) { return 1 << (FWD ? 4 - ilog2c(s + 1) : ilog2c(s + 1)); }

// ---- LDS word addresses of the four register layouts ------------------------------------------
// (all already swizzled: addr(i) = i ^ ((i >> 5) & 31))
//
//   layout A ("strided"):   r = (kk << PB) | pp,  i = (kk << 10) | (tid << PB) | pp
//   layout B ("blocked"):   r = m,                i = (blk << 10) | (m << 5) | j,  blk = tid >> 5, j = tid & 31
//   layout C ("contiguous"): r,                   i = (tid << 5) | r
//   layout S ("stream"):    r = (s << 1) | e,     i = (wave << 11) | (s << 7) | (lane << 1) | e
//                           (a lane moves 16 contiguous bytes, a wave 1 KiB per HBM instruction)
// N = 32768 uses a PADDED buffer instead (word address i + (i >> 5), 132 KiB): every register's offset is then an immediate of the
// ds instruction and the 279 v_xor per thread disappear; it is conflict-free for all four layouts at that size only (at N <= 16384
// the strided layout A collides, and the padding would cost the second workgroup per CU its LDS).
// (forward kernels only: the inverse kernel with the padded buffer spills 34 registers and runs 4 % slower)
template <int LOGN, bool PADDED = false> struct Addr {
    using G = Geo<LOGN>;
    static constexpr int LOGN_ = LOGN;
    static constexpr bool PAD = PADDED;
    static constexpr int WORDS = PAD ? G::N + G::N / 32 : G::N;
    u32 a_base;   // ((tid << PB) ^ h) with h = bits 9..5 of (tid << PB)
    u32 b_base;   // (blk << 10) | j
    u32 c_base;   // (tid << 5) | (tid & 31)
    u32 s_base;   // (wave << 11) | ((lane >> 4) << 5) | (((lane & 15) << 1) ^ (lane >> 4))
    HP_DEV void init(u32 tid) {
        const u32 lane = tid & 63u, wave = tid >> 6;
        if (PAD) {   // byte offsets of i + (i >> 5) with the register-dependent part left to lay_addr()
            a_base = ((tid << G::PB) + ((tid << G::PB) >> 5)) << 2;   // (the passenger bits pp < 2^PB <= 32 do not reach bit 5)
            b_base = ((tid >> 5) * 1056u + (tid & 31u)) << 2;
            c_base = (tid * 33u) << 2;
            s_base = (wave * 2112u + 2u * lane + (lane >> 4)) << 2;
            return;
        }
        // all four are BYTE offsets into the exchange buffer (word index * 4): an access then costs one v_xor, the
        // additive part of lay_addr() folds into the ds instruction's immediate offset
        const u32 t = tid << G::PB;
        a_base = (t ^ ((t >> 5) & 31u)) << 2;
        b_base = (((tid >> 5) << 10) | (tid & 31u)) << 2;
        c_base = ((tid << 5) | (tid & 31u)) << 2;
        s_base = ((wave << 11) | ((lane >> 4) << 5) | (((lane & 15u) << 1) ^ (lane >> 4))) << 2;
    }
};

enum { LAY_A = 0, LAY_B = 1, LAY_C = 2, LAY_S = 3 };

// Launder a thread-constant through an empty asm so the compiler treats it as a fresh value:
// the 32 LDS addresses derived from it are then recomputed (one v_xor each) in every round of
// an exchange instead of being kept live in 32 VGPRs across rounds.
HP_DEV u32 opaque(u32 v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int LAY, class AD> HP_DEV u32 lay_base(const AD &ad) {
    if (LAY == LAY_A) return ad.a_base;
    if (LAY == LAY_B) return ad.b_base;
    if (LAY == LAY_C) return ad.c_base;
    return ad.s_base;
}

template <int LOGN, int LAY, bool PAD> HP_DEV u32 lay_addr(u32 base, int r) {
    using G = Geo<LOGN>;
    if (PAD) {
        if (LAY == LAY_A) return base + ((u32)(r >> G::PB) * 1056u + (u32)(r & ((1 << G::PB) - 1))) * 4u;   // i = (kk << 10) | (tid << PB) | pp
        if (LAY == LAY_B) return base + (u32)r * (33u * 4u);                         // i = (blk << 10) | (r << 5) | j
        if (LAY == LAY_C) return base + (u32)r * 4u;                                 // i = (tid << 5) | r
        return base + ((u32)(r >> 1) * 132u + (u32)(r & 1)) * 4u;                    // i = (wave << 11) | (s << 7) | (lane << 1) | e
    }
    if (LAY == LAY_A) return (base ^ ((u32)(r & ((1 << G::PB) - 1)) << 2)) + ((u32)((r >> G::PB) << 10) << 2);
    if (LAY == LAY_B) return (base ^ ((u32)r << 2)) + ((u32)(r << 5) << 2);
    if (LAY == LAY_C) return base ^ ((u32)r << 2);
    return (base ^ ((u32)((r & 1) | (((r >> 1) & 7) << 2)) << 2)) + ((u32)((r >> 1) << 7) << 2);
}

// Between the write and the read phase of an exchange round.  Workgroup-wide: s_barrier.  Wave-local: the hardware
// executes one wave's LDS instructions in order, so no wait is needed -- but the COMPILER must not move a thread's reads
// above its own writes (different addresses for the thread, the same words for its wave): memory clobber + scheduling
// barrier.  Measured cost: none.
template <bool WG> HP_DEV void exch_fence() {
    if (WG) {
        __syncthreads();
    } else {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// word of the exchange buffer at a byte offset
HP_DEV u32 &lds_w(u32 *lds, u32 byte_off) { return *reinterpret_cast<u32 *>(reinterpret_cast<char *>(lds) + byte_off); }

// Transpose the workgroup's coefficients from register layout FROM to layout TO through LDS, one
// 32-bit half at a time.  WG: the exchange crosses waves (needs s_barrier); otherwise it is
// confined to the wave's own 2048-word region and relies on in-order LDS execution per wave.
template <int LOGN, int FROM, int TO, bool WG, class AD>
HP_DEV void exchange(u64 (&x)[32], u32 *lds, const AD &ad) {
    static_assert(AD::LOGN_ == LOGN, "address set of another ring degree");
    constexpr bool PAD = AD::PAD;
    u32 keep[32];
    {
        const u32 fb = opaque(lay_base<FROM>(ad));
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            lds_w(lds, lay_addr<LOGN, FROM, PAD>(fb, r)) = lo32(x[r]);
            keep[r] = hi32(x[r]);
        }
    }
    exch_fence<WG>();
    u32 nlo[32];
    {
        const u32 tb = opaque(lay_base<TO>(ad));
#pragma unroll
        for (int r = 0; r < 32; ++r) nlo[r] = lds_w(lds, lay_addr<LOGN, TO, PAD>(tb, r));
    }
    exch_fence<WG>();
    {
        const u32 fb = opaque(lay_base<FROM>(ad));
#pragma unroll
        for (int r = 0; r < 32; ++r) lds_w(lds, lay_addr<LOGN, FROM, PAD>(fb, r)) = keep[r];
    }
    exch_fence<WG>();
    {
        const u32 tb = opaque(lay_base<TO>(ad));
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = mk64(nlo[r], lds_w(lds, lay_addr<LOGN, TO, PAD>(tb, r)));
    }
}

constexpr int HP_EPI_DEPTH = 4;   // rows of the fused drop epilogue whose operand loads are in flight

struct alignas(16) V2 {
    u64 x, y;
};

// Streaming accesses to coefficient data are marked non-temporal so that the once-read, once-written
// limbs do not push the twiddle tables (re-read by every workgroup) out of L2: +3.5..5 % on every
// transform shape (HP_TEMPORAL_DATA restores plain accesses for A/B runs).
HP_DEV V2 ld_stream(const u64 *p) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    const vv v = __builtin_nontemporal_load(reinterpret_cast<const vv *>(p));
    return V2{v.x, v.y};
}
HP_DEV void st_stream(u64 *p, const V2 &v) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    __builtin_nontemporal_store(vv{v.x, v.y}, reinterpret_cast<vv *>(p));
}

// A coefficient row as a buffer: 16 bytes per lane at (lane byte offset, wave-uniform byte offset), non-temporal like ld_stream / st_stream
struct StreamBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    // the row address is the same for the whole workgroup; say so (readfirstlane), or the compiler keeps the descriptor in VGPRs
    // and wraps every access in a loop over its distinct values
    HP_DEV static u64 *uniform(const u64 *row) {
        const u64 a = reinterpret_cast<u64>(row);
        const u32 lo = (u32)__builtin_amdgcn_readfirstlane((u32)a), hi = (u32)__builtin_amdgcn_readfirstlane((u32)(a >> 32));   // (the builtin returns int)
        return reinterpret_cast<u64 *>(((u64)hi << 32) | lo);
    }
    HP_DEV explicit StreamBuf(const u64 *row) : rsrc(__builtin_amdgcn_make_buffer_rsrc(uniform(row), 0, 0x7fffffff, 0x00020000)) {}
    HP_DEV V2 load(u32 voff, u32 soff) const {
        const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 2);
        return V2{((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
    }
    // Stores take the row offset in the LANE offset and no scalar offset.  With an SGPR soffset the assembler assumes that a
    // 16-byte buffer store has read its data by the time the next instruction issues and pads nothing; on gfx950 it has not: a
    // VALU instruction right behind the store that overwrites a data register changed what the last lanes of each 16-lane pass
    // stored (found as rare wrong words in lanes 12-15, 28-31, ... of one row).  Without soffset the hazard is known and padded.
    HP_DEV void store(u32 voff, const V2 &v) const {
        __builtin_amdgcn_raw_buffer_store_b128(v4u32{(u32)v.x, (u32)(v.x >> 32), (u32)v.y, (u32)(v.y >> 32)}, rsrc, voff, 0, 2);
    }
};

// value held by the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2], no LDS involved
HP_DEV u64 from_pair_lane(u64 v) {
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0xB1, 0xF, 0xF, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0xB1, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}

// load-side work deferred into the first stage of a forward pass (see pass_slots): the lane-pair swap of N = 32768
HP_DEV void lazy_swap(u64 (&x)[32], int r) {
    const bool odd = (threadIdx.x & 1u) != 0;
    const u64 vx = x[r], vy = x[r + 1];
    const u64 keep = odd ? vy : vx, send = odd ? vx : vy;
    const u64 recv = from_pair_lane(send);
    x[r] = odd ? recv : keep;
    x[r + 1] = odd ? keep : recv;
}
struct NoPre {
    static constexpr bool on = false;
    HP_DEV void operator()(u64 (&)[32], int) const {}
};
struct SwapPre {
    static constexpr bool on = true;
    HP_DEV void operator()(u64 (&x)[32], int r) const { lazy_swap(x, r); }
};

// ---- forward kernel ----------------------------------------------------------------------------
#define HP_LOAD_ORDER(t) (((t) >> 1) | (((t) & 1) << 3))
// load, layout A: thread reads 2^PB consecutive coefficients at 2^A places 1024 apart
template <int LOGN, bool LZ = false>
HP_DEV void load_flight(const u64 *src, u32 tid, u64 (&x)[32]) {
    using G = Geo<LOGN>;
    const u64 *s = src + ((size_t)tid << G::PB);
    if (G::PB == 0) {
        // N = 32768: a thread owns one column (tid) of 32 rows 1024 apart.  Two neighbouring lanes
        // fetch 16 bytes (both their columns) of alternate rows and trade halves with one DPP swap,
        // so every HBM instruction still moves 16 bytes per lane.
        const bool odd = (tid & 1u) != 0;
        const u64 *sp = src + (tid & ~1u);
        // issue order 0, 8, 1, 9, ...: the first stage pairs register r with r + 16, i.e. load p with load p + 8, so its
        // butterflies can start as soon as the first two loads are back instead of after nine
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int p = HP_LOAD_ORDER(t);
            const V2 v = ld_stream(sp + ((size_t)(2 * p + (odd ? 1 : 0)) << 10));
            if (LZ) { x[2 * p] = v.x; x[2 * p + 1] = v.y; continue; }   // sorted by lazy_swap in the first pass
            const u64 keep = odd ? v.y : v.x, send = odd ? v.x : v.y;
            const u64 recv = from_pair_lane(send);
            x[2 * p] = odd ? recv : keep;
            x[2 * p + 1] = odd ? keep : recv;
        }
    }
#pragma unroll
    for (int tk = 0; tk < (G::PB == 0 ? 0 : (1 << G::A)); ++tk) {
        // same idea as above: the first stage pairs place kk with kk + 2^(A-1)
        const int kk = (tk >> 1) | ((tk & 1) << (G::A - 1));
        if (G::PB == 0) {
        } else {
#pragma unroll
            for (int pp = 0; pp < (1 << G::PB); pp += 2) {
                const V2 v = ld_stream(s + ((size_t)kk << 10) + pp);
                x[(kk << G::PB) | pp] = v.x;
                x[(kk << G::PB) | pp | 1] = v.y;
            }
        }
    }
}

// Phase tracing for kernel tuning (the including .hip defines `__device__ u64 g_trace[2 * 2048 * 16 * HP_TRACE_SLOTS]` and exports a reader)
// Phase tracing for kernel tuning (variant builds only, -DHP_TRACE): shader-clock stamps of wave 0
// of every 16th workgroup at the phase boundaries, read back with hp_debug_trace().
#ifdef HP_TRACE
#ifdef HP_TRACE_ALL
#define HP_TRACE_SEL (blockIdx.x < 2048 * 16)
#define HP_TRACE_IDX (blockIdx.x)
#else
#define HP_TRACE_SEL ((blockIdx.x & 15) == 0 && (blockIdx.x >> 4) < 2048)
#define HP_TRACE_IDX (blockIdx.x >> 4)
#endif
#define HP_TRACE_SLOTS 12
#define TRACE_DECL u64 tr__[HP_TRACE_SLOTS]; int tri__ = 0; tr__[10] = ((u64)__builtin_amdgcn_s_getreg(63492) << 32) | (u32)__builtin_amdgcn_s_getreg((31 << 11) | 20); tr__[11] = t_entry__;
#define TRACE_ENTRY __builtin_amdgcn_sched_barrier(0); const u64 t_entry__ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define TRACE_MARK() do { __builtin_amdgcn_sched_barrier(0); tr__[tri__++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef HP_TRACE_WAVES   // every wave of every 16th workgroup: record (blockIdx >> 4) * 16 + wave
#define TRACE_FLUSH() do { if ((threadIdx.x & 63u) == 0 && HP_TRACE_SEL && (threadIdx.x >> 6) < 16) { \
        for (int i__ = 0; i__ < HP_TRACE_SLOTS; i__++) g_trace[(HP_TRACE_IDX * 16 + (threadIdx.x >> 6)) * HP_TRACE_SLOTS + i__] = (i__ < tri__ || i__ >= 10) ? tr__[i__] : 0; } } while (0)
#else
#define TRACE_FLUSH() do { if ((threadIdx.x == 0 || threadIdx.x == blockDim.x - 64) && HP_TRACE_SEL) { \
        for (int i__ = 0; i__ < HP_TRACE_SLOTS; i__++) g_trace[(HP_TRACE_IDX * 2 + (threadIdx.x != 0)) * HP_TRACE_SLOTS + i__] = (i__ < tri__ || i__ >= 10) ? tr__[i__] : 0; } } while (0)
#endif
#else
#define TRACE_DECL
#define TRACE_ENTRY
#define TRACE_MARK() do { } while (0)
#define TRACE_FLUSH() do { } while (0)
#endif

} // namespace
