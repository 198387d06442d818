// hp_elem.hip -- coefficient-wise CDNA4 kernels (HBM-streaming, 16-byte accesses).
//
// Bound: HBM.  Algorithmic bytes per limb of n words: binary op 24n, unary 16n
// (SURVEY.md section 8d).  One workgroup streams one 2048-word chunk of one
// limb with 16-byte loads/stores; the limb's constants are wave-uniform and
// come from scalar loads of the plan entry.
#include "hp_kernels.h"
#include <cstdlib>

#define ELEM_THREADS 256
#define ELEM_CHUNK 2048u   // words per workgroup = 256 threads x 4 x 16 B

struct alignas(16) U2 {
    u64 x, y;
};

// streaming accesses (read once / written once per launch, far more data than L2 holds): non-temporal
HP_DEV U2 ld_nt(const u64 *p) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    const vv v = __builtin_nontemporal_load(reinterpret_cast<const vv *>(p));
    return U2{v.x, v.y};
}
HP_DEV void st_nt(u64 *p, const U2 &v) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    __builtin_nontemporal_store(vv{v.x, v.y}, reinterpret_cast<vv *>(p));
}

static inline void elem_grid(u32 n, u32 rows, u32 &chunks, dim3 &grid) {
    chunks = (n + ELEM_CHUNK - 1) / ELEM_CHUNK;
    grid = dim3(chunks * rows, 1, 1);
}

// ---- binary: rns.cpp:58-87 (add), :89-118 (sub), :120-140 (mul) ----------------
template <int OP>
__global__ void __launch_bounds__(ELEM_THREADS) k_poly_binary(const HpLimb *__restrict__ limbs, u32 L, u32 n,
                                                             u32 chunks, const u64 *a,
                                                             const u64 *b, u64 *out) {   // out may be a (operator+=)
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const HpLimb m = limbs[row % L];
    const size_t base = (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    if ((chunk + 1) * ELEM_CHUNK <= n) {
        // a full chunk (every chunk of the tiled ring degrees): the eight 16-byte loads of a thread are issued together, then
        // the arithmetic and the four stores -- the launch shape a plain copy streams fastest with (tools/ubench/ubench_copy.hip)
        constexpr int IT = ELEM_CHUNK / (ELEM_THREADS * 2);
        const size_t o = base + (size_t)chunk * ELEM_CHUNK + threadIdx.x * 2;
        U2 va[IT], vb[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) { va[t] = ld_nt(a + o + (size_t)t * ELEM_THREADS * 2); vb[t] = ld_nt(b + o + (size_t)t * ELEM_THREADS * 2); }
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            U2 r;
            if (OP == HP_ADD) { r.x = hp_add_lazy(va[t].x, vb[t].x, m.two_q); r.y = hp_add_lazy(va[t].y, vb[t].y, m.two_q); }
            if (OP == HP_SUB) { r.x = hp_sub_lazy(va[t].x, vb[t].x, m.two_q); r.y = hp_sub_lazy(va[t].y, vb[t].y, m.two_q); }
            if (OP == HP_MUL) { r.x = hp_mul_hybrid_lazy(va[t].x, vb[t].x, m); r.y = hp_mul_hybrid_lazy(va[t].y, vb[t].y, m); }
            st_nt(out + o + (size_t)t * ELEM_THREADS * 2, r);
        }
        return;
    }
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x * 2; i < end; i += ELEM_THREADS * 2) {
        if (i + 1 < end) {
            U2 va = ld_nt(a + base + i);
            U2 vb = ld_nt(b + base + i);
            U2 r;
            if (OP == HP_ADD) { r.x = hp_add_lazy(va.x, vb.x, m.two_q); r.y = hp_add_lazy(va.y, vb.y, m.two_q); }
            if (OP == HP_SUB) { r.x = hp_sub_lazy(va.x, vb.x, m.two_q); r.y = hp_sub_lazy(va.y, vb.y, m.two_q); }
            if (OP == HP_MUL) { r.x = hp_mul_hybrid_lazy(va.x, vb.x, m); r.y = hp_mul_hybrid_lazy(va.y, vb.y, m); }
            st_nt(out + base + i, r);
        } else {
            u64 va = a[base + i], vb = b[base + i], r = 0;
            if (OP == HP_ADD) r = hp_add_lazy(va, vb, m.two_q);
            if (OP == HP_SUB) r = hp_sub_lazy(va, vb, m.two_q);
            if (OP == HP_MUL) r = hp_mul_hybrid_lazy(va, vb, m);
            out[base + i] = r;
        }
    }
}

hipError_t hp_launch_poly_binary(int op, const HpLimb *limbs, u32 L, u32 n, u32 rows, const u64 *a,
                                 const u64 *b, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, rows, chunks, grid);
    if (op == HP_ADD) k_poly_binary<HP_ADD><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, a, b, out);
    else if (op == HP_SUB) k_poly_binary<HP_SUB><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, a, b, out);
    else k_poly_binary<HP_MUL><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, a, b, out);
    return hipGetLastError();
}

// ---- a chain of += / -= in one pass: rns.cpp:58-118 term after term ----------------------------------------------------
// out[p] = ((x_0 op_1 x_1) op_2 x_2) ... of polynomial p's `terms` operand rows (anywhere; addresses as kernel arguments), op_j = -= where
// bit j of neg is set: the words of the single calls in their order (each step is the lazy add / sub of rns.cpp), the intermediate
// sums never cross HBM.  The accumulate of a diagonal loop (src/circuits/linear_algebra.h:117-121) is such a chain.
__global__ void __launch_bounds__(ELEM_THREADS) k_poly_fold(const HpLimb *__restrict__ limbs, u32 L, u32 n, u32 chunks, u32 terms,
                                                           HpFoldRows rows, u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p * L + k
    const u32 p = row / L, k = row % L;
    const u64 two_q = limbs[k].two_q;
    const size_t off = (size_t)k * n;
    const u64 *const *src = rows.p + (size_t)p * terms;
    u64 *dst = out + (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    if ((chunk + 1) * ELEM_CHUNK <= n) {   // a full chunk: four 16-byte loads of a term in flight per thread
        constexpr int IT = ELEM_CHUNK / (ELEM_THREADS * 2);
        const size_t o = off + (size_t)chunk * ELEM_CHUNK + threadIdx.x * 2;
        U2 acc[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) acc[t] = ld_nt(src[0] + o + (size_t)t * ELEM_THREADS * 2);
        for (u32 j = 1; j < terms; ++j) {
            const u64 *x = src[j] + o;
            U2 v[IT];
#pragma unroll
            for (int t = 0; t < IT; ++t) v[t] = ld_nt(x + (size_t)t * ELEM_THREADS * 2);
            if ((rows.neg >> j) & 1u) {
#pragma unroll
                for (int t = 0; t < IT; ++t) { acc[t].x = hp_sub_lazy(acc[t].x, v[t].x, two_q); acc[t].y = hp_sub_lazy(acc[t].y, v[t].y, two_q); }
            } else {
#pragma unroll
                for (int t = 0; t < IT; ++t) { acc[t].x = hp_add_lazy(acc[t].x, v[t].x, two_q); acc[t].y = hp_add_lazy(acc[t].y, v[t].y, two_q); }
            }
        }
#pragma unroll
        for (int t = 0; t < IT; ++t) st_nt(dst + (o - off) + (size_t)t * ELEM_THREADS * 2, acc[t]);
        return;
    }
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        u64 acc = src[0][off + i];
        for (u32 j = 1; j < terms; ++j) {
            const u64 v = src[j][off + i];
            acc = ((rows.neg >> j) & 1u) ? hp_sub_lazy(acc, v, two_q) : hp_add_lazy(acc, v, two_q);
        }
        dst[i] = acc;
    }
}

hipError_t hp_launch_poly_fold(const HpLimb *limbs, u32 L, u32 n, u32 polys, u32 terms, const HpFoldRows &rows, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, polys * L, chunks, grid);
    k_poly_fold<<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, terms, rows, out);
    return hipGetLastError();
}

// ---- unary: rns.cpp:142-171 (scalar multiply), mod_arith.h:65-72 (strict) --------
template <int STRICT>
__global__ void __launch_bounds__(ELEM_THREADS) k_poly_unary(const HpLimb *__restrict__ limbs, HpScalars sc, u32 L,
                                                            u32 n, u32 chunks, const u64 *a,
                                                            u64 *out) {   // in-place use: out == a
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u32 k = row % L;
    const u64 q = limbs[k].q;
    const u64 s = sc.s[k], sh = sc.sh[k];
    const size_t base = (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    if ((chunk + 1) * ELEM_CHUNK <= n) {   // a full chunk: all loads of the thread first (see k_poly_binary)
        constexpr int IT = ELEM_CHUNK / (ELEM_THREADS * 2);
        const size_t o = base + (size_t)chunk * ELEM_CHUNK + threadIdx.x * 2;
        U2 v[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) v[t] = ld_nt(a + o + (size_t)t * ELEM_THREADS * 2);
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            if (STRICT) { v[t].x = hp_strict(v[t].x, q); v[t].y = hp_strict(v[t].y, q); }
            else { v[t].x = hp_harvey_lazy(v[t].x, s, sh, q); v[t].y = hp_harvey_lazy(v[t].y, s, sh, q); }
            st_nt(out + o + (size_t)t * ELEM_THREADS * 2, v[t]);
        }
        return;
    }
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x * 2; i < end; i += ELEM_THREADS * 2) {
        if (i + 1 < end) {
            U2 v = ld_nt(a + base + i);
            if (STRICT) { v.x = hp_strict(v.x, q); v.y = hp_strict(v.y, q); }
            else { v.x = hp_harvey_lazy(v.x, s, sh, q); v.y = hp_harvey_lazy(v.y, s, sh, q); }
            st_nt(out + base + i, v);
        } else {
            u64 v = a[base + i];
            out[base + i] = STRICT ? hp_strict(v, q) : hp_harvey_lazy(v, s, sh, q);
        }
    }
}

hipError_t hp_launch_poly_scalar_mul(const HpLimb *limbs, const HpScalars &sc, u32 L, u32 n, u32 rows,
                                     const u64 *a, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, rows, chunks, grid);
    k_poly_unary<0><<<grid, ELEM_THREADS, 0, stream>>>(limbs, sc, L, n, chunks, a, out);
    return hipGetLastError();
}

hipError_t hp_launch_poly_strict(const HpLimb *limbs, u32 L, u32 n, u32 rows, u64 *x, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, rows, chunks, grid);
    HpScalars sc = {};
    k_poly_unary<1><<<grid, ELEM_THREADS, 0, stream>>>(limbs, sc, L, n, chunks, x, x);
    return hipGetLastError();
}

// ---- deep copy of device words: allocator.h:113-118 (SmartArray's copy constructor) for limbs that live in HBM ----------
// Also the engine's measured stream ceiling (bench.py "hbm_copy_ceiling_GBps"): one read and one write stream, nothing else.
#define COPY_UNROLL 4
__global__ void __launch_bounds__(ELEM_THREADS) k_copy(size_t pairs, const u64 *__restrict__ in, u64 *__restrict__ out) {
    // a workgroup moves COPY_UNROLL x 4 KiB; the loads of a round are issued together
    const size_t base = (size_t)blockIdx.x * (ELEM_THREADS * COPY_UNROLL) + threadIdx.x;
    U2 v[COPY_UNROLL];
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const size_t i = base + (size_t)u * ELEM_THREADS;
        if (i < pairs) v[u] = ld_nt(in + 2 * i);
    }
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; ++u) {
        const size_t i = base + (size_t)u * ELEM_THREADS;
        if (i < pairs) st_nt(out + 2 * i, v[u]);
    }
}

hipError_t hp_launch_copy(size_t words, const u64 *in, u64 *out, hipStream_t stream) {
    const size_t pairs = words >> 1;
    if (pairs) {
        const size_t per = (size_t)ELEM_THREADS * COPY_UNROLL;
        k_copy<<<dim3((unsigned)((pairs + per - 1) / per)), ELEM_THREADS, 0, stream>>>(pairs, in, out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (words & 1) return hipMemcpyAsync(out + words - 1, in + words - 1, 8, hipMemcpyDeviceToDevice, stream);
    return hipSuccess;
}

// ---- rows that live in separate blocks of registered HOST memory (hehub's SmartArray limbs, allocator.h:105-223) ------------------
// One kernel moves a whole polynomial between its contiguous device rows and its scattered host blocks through their device-visible
// addresses: 47-49 GB/s over PCIe either way against 11-17 GB/s for one DMA command per 256 KiB block (tools/ubench/ubench_pcie.hip).
template <bool TO_HOST> __global__ void __launch_bounds__(256) k_host_rows(HpHostRows rows, u64 *dev, size_t pairs) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    vv *d = reinterpret_cast<vv *>(dev) + (size_t)blockIdx.y * pairs;
    vv *h = reinterpret_cast<vv *>(rows.p[blockIdx.y]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        if (TO_HOST) h[i] = d[i];
        else d[i] = h[i];
    }
}
hipError_t hp_launch_host_rows(bool to_host, const HpHostRows &rows, u32 count, size_t words, u64 *dev, hipStream_t stream) {
    if (count == 0 || words == 0) return hipSuccess;
    const size_t pairs = words >> 1;
    const unsigned gx = (unsigned)((pairs + 256 * 16 - 1) / (256 * 16));   // 16 pairs per thread
    if (to_host) k_host_rows<true><<<dim3(gx ? gx : 1, count), 256, 0, stream>>>(rows, dev, pairs);
    else k_host_rows<false><<<dim3(gx ? gx : 1, count), 256, 0, stream>>>(rows, dev, pairs);
    return hipGetLastError();
}

// ---- gathers: permutation.cpp:28-75 -------------------------------------------------
__global__ void __launch_bounds__(ELEM_THREADS) k_gather(const u32 *__restrict__ perm, u32 n, u32 chunks,
                                                        const u64 *__restrict__ in, u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const size_t base = (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    // (the gathered row is re-read line by line and wants the caches; the output is written once)
    if ((chunk + 1) * ELEM_CHUNK <= n) {   // a full chunk: the eight index loads, then the eight gathers, in flight together
        constexpr int IT = ELEM_CHUNK / ELEM_THREADS;
        const u32 i0 = chunk * ELEM_CHUNK + threadIdx.x;
        u32 idx[IT];
        u64 v[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) idx[t] = perm[i0 + t * ELEM_THREADS];
#pragma unroll
        for (int t = 0; t < IT; ++t) v[t] = in[base + idx[t]];
#pragma unroll
        for (int t = 0; t < IT; ++t) __builtin_nontemporal_store(v[t], out + base + i0 + t * ELEM_THREADS);
        return;
    }
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) __builtin_nontemporal_store(in[base + perm[i]], out + base + i);
}

hipError_t hp_launch_gather(const u32 *perm, u32 n, u32 rows, const u64 *in, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, rows, chunks, grid);
    k_gather<<<grid, ELEM_THREADS, 0, stream>>>(perm, n, chunks, in, out);
    return hipGetLastError();
}

__global__ void __launch_bounds__(ELEM_THREADS) k_reverse(u32 n, u32 chunks, const u64 *__restrict__ in,
                                                         u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const size_t base = (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) out[base + i] = in[base + (n - 1 - i)];
}

// cycle / involution of several ciphertexts in one launch, each with its own map and its two polynomials anywhere in device memory
// (hp_dev_ckks_rotate_many: the sources and maps travel as kernel arguments): out u64[count][2][L][N]
__global__ void __launch_bounds__(ELEM_THREADS) k_gather_many(HpGatherTable tab, u32 n, u32 chunks, u32 L, u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u32 b = row / (2 * L), r = row % (2 * L), h = r / L, k = r % L;
    const u64 *__restrict__ in = tab.src[b][h] + (size_t)k * n;
    const u32 *__restrict__ perm = tab.perm[b];
    u64 *__restrict__ o = out + (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    if (!perm) {   // involution: permutation.cpp:57-75
        for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) __builtin_nontemporal_store(in[n - 1 - i], o + i);
        return;
    }
    if ((chunk + 1) * ELEM_CHUNK <= n) {
        constexpr int IT = ELEM_CHUNK / ELEM_THREADS;
        const u32 i0 = chunk * ELEM_CHUNK + threadIdx.x;
        u32 idx[IT];
        u64 v[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) idx[t] = perm[i0 + t * ELEM_THREADS];
#pragma unroll
        for (int t = 0; t < IT; ++t) v[t] = in[idx[t]];
#pragma unroll
        for (int t = 0; t < IT; ++t) __builtin_nontemporal_store(v[t], o + i0 + t * ELEM_THREADS);
        return;
    }
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) __builtin_nontemporal_store(in[perm[i]], o + i);
}

hipError_t hp_launch_gather_many(const HpGatherTable &tab, u32 count, u32 n, u32 L, u64 *out, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (count > HP_GATHER_TABLE_MAX) return hipErrorInvalidValue;
    u32 chunks; dim3 grid;
    elem_grid(n, count * 2 * L, chunks, grid);
    k_gather_many<<<grid, ELEM_THREADS, 0, stream>>>(tab, n, chunks, L, out);
    return hipGetLastError();
}

hipError_t hp_launch_reverse(u32 n, u32 rows, const u64 *in, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, rows, chunks, grid);
    k_reverse<<<grid, ELEM_THREADS, 0, stream>>>(n, chunks, in, out);
    return hipGetLastError();
}

// ---- single-vector kernels (drop-in mod_arith entry points) --------------------------
template <int OP>
__global__ void __launch_bounds__(ELEM_THREADS) k_vec(HpVecConsts c, size_t n, const u64 *__restrict__ a,
                                                     const u64 *__restrict__ b, u64 *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * ELEM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * ELEM_THREADS) {
        u64 r = 0;
        if (OP == HP_V_BARRETT_LAZY) r = hp_barrett_lazy(a[i], c.q, c.barrett_c);            // mod_arith.cpp:9-17
        if (OP == HP_V_BARRETT) r = hp_strict(hp_barrett_lazy(a[i], c.q, c.barrett_c), c.q);   // mod_arith.h:18-25
        if (OP == HP_V_STRICT) r = hp_strict(a[i], c.q);                                       // mod_arith.h:58-63
        if (OP == HP_V_MUL_HYBRID) {                                                           // mod_arith.cpp:64-92
            u64 lo, hi;
            hp_mul128(a[i], b[i], lo, hi);
            u64 t = hp_montgomery128_lazy(lo, hi, c.q, c.mqinv);
            r = hp_harvey_lazy(t, c.r64, c.r64h, c.q);
        }
        if (OP == HP_V_MUL_BARRETT) {                                                          // mod_arith.cpp:94-111
            u64 al, ah;
            hp_mul128(a[i], b[i], al, ah);
            // approx_quotient = ah*ch + ((ah*cl + al*ch) >> 64)   (the inner sum is a wrapping u128)
            u64 p1l, p1h, p2l, p2h;
            hp_mul128(ah, c.c128_lo, p1l, p1h);
            hp_mul128(al, c.c128_hi, p2l, p2h);
            u64 sl = p1l + p2l;
            u64 sh = p1h + p2h + (sl < p1l ? 1ull : 0ull);
            u64 qhat = ah * c.c128_hi + sh;
            r = al - c.q * qhat;
        }
        if (OP == HP_V_MONTGOMERY128) r = hp_montgomery128_lazy(a[2 * i], a[2 * i + 1], c.q, c.mqinv);
        out[i] = r;
    }
}

hipError_t hp_launch_vec(int op, const HpVecConsts &c, size_t n, const u64 *a, const u64 *b, u64 *out,
                         hipStream_t stream) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + ELEM_THREADS - 1) / ELEM_THREADS;
    if (blocks > 256 * 16) blocks = 256 * 16;
    dim3 grid((unsigned)blocks);
    switch (op) {
    case HP_V_BARRETT_LAZY: k_vec<HP_V_BARRETT_LAZY><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    case HP_V_BARRETT: k_vec<HP_V_BARRETT><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    case HP_V_STRICT: k_vec<HP_V_STRICT><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    case HP_V_MUL_HYBRID: k_vec<HP_V_MUL_HYBRID><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    case HP_V_MUL_BARRETT: k_vec<HP_V_MUL_BARRETT><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    case HP_V_MONTGOMERY128: k_vec<HP_V_MONTGOMERY128><<<grid, ELEM_THREADS, 0, stream>>>(c, n, a, b, out); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- fused tensor product: ckks/arith.cpp:55-62 / bgv/arith.cpp:59-69 -------------------
// d0 = a0*b0, d1 = (a0*b1) + (a1*b0), d2 = a1*b1.  Reads 4 limbs, writes 3: 56n bytes per limb index.
// ROWS: the four operand polynomials of ciphertext pair p by address (an application's ciphertexts are separate objects: the fused
// pipelines read them where they lie, hp_dev_*_mult_*_rows); the addresses travel as kernel arguments
template <bool ROWS>
__global__ void __launch_bounds__(ELEM_THREADS) k_tensor(const HpLimb *__restrict__ limbs, u32 L, u32 k_first, u32 kc,
                                                        u32 n, u32 chunks, u32 cw, const u64 *__restrict__ ct1,
                                                        const u64 *__restrict__ ct2, HpTensorRows rows, u64 *__restrict__ quad) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*kc + (k - k_first); cw = words per workgroup
    const u32 p = row / kc, k = k_first + row % kc;
    const HpLimb m = limbs[k];
    const size_t poly = (size_t)L * n;
    const u64 *a0 = ROWS ? rows.p[p][0] + (size_t)k * n : ct1 + (size_t)p * 2 * poly + (size_t)k * n;
    const u64 *a1 = ROWS ? rows.p[p][1] + (size_t)k * n : a0 + poly;
    const u64 *b0 = ROWS ? rows.p[p][2] + (size_t)k * n : ct2 + (size_t)p * 2 * poly + (size_t)k * n;
    const u64 *b1 = ROWS ? rows.p[p][3] + (size_t)k * n : b0 + poly;
    u64 *d0 = quad + (size_t)p * 3 * poly + (size_t)k * n, *d1 = d0 + poly, *d2 = d1 + poly;
    const u32 end = min(n, (chunk + 1) * cw);
    // (issuing the loads of several steps together, which gains 5-9 % in k_poly_binary, measured +-0 here: 0.830 vs 0.827 ms)
    for (u32 i = chunk * cw + threadIdx.x * 2; i < end; i += ELEM_THREADS * 2) {
        if (i + 1 < end) {
            U2 va0 = ld_nt(a0 + i), va1 = ld_nt(a1 + i);
            U2 vb0 = ld_nt(b0 + i), vb1 = ld_nt(b1 + i);
            U2 r0, r1, r2;
            r0.x = hp_mul_hybrid_lazy(va0.x, vb0.x, m);
            r0.y = hp_mul_hybrid_lazy(va0.y, vb0.y, m);
            r1.x = hp_add_lazy(hp_mul_hybrid_lazy(va0.x, vb1.x, m), hp_mul_hybrid_lazy(va1.x, vb0.x, m), m.two_q);
            r1.y = hp_add_lazy(hp_mul_hybrid_lazy(va0.y, vb1.y, m), hp_mul_hybrid_lazy(va1.y, vb0.y, m), m.two_q);
            r2.x = hp_mul_hybrid_lazy(va1.x, vb1.x, m);
            r2.y = hp_mul_hybrid_lazy(va1.y, vb1.y, m);
            st_nt(d0 + i, r0);
            st_nt(d1 + i, r1);
            st_nt(d2 + i, r2);
        } else {
            u64 x0 = a0[i], x1 = a1[i], y0 = b0[i], y1 = b1[i];
            d0[i] = hp_mul_hybrid_lazy(x0, y0, m);
            d1[i] = hp_add_lazy(hp_mul_hybrid_lazy(x0, y1, m), hp_mul_hybrid_lazy(x1, y0, m), m.two_q);
            d2[i] = hp_mul_hybrid_lazy(x1, y1, m);
        }
    }
}

// a workgroup covers 2048 words of a limb in four dependent load -> multiply -> store steps per thread; a launch of a few limbs (one
// ciphertext through hehub's one-call-per-ciphertext interface: 160 workgroups at C3) is then four memory latencies long with a
// third of the CUs idle -- such a launch gets 512 words per workgroup (one step per thread)
static inline void tensor_grid(u32 n, u32 rows, u32 &chunks, u32 &cw, dim3 &grid) {
    cw = ELEM_CHUNK;
    if ((size_t)rows * ((n + ELEM_CHUNK - 1) / ELEM_CHUNK) < 1024 && n >= ELEM_THREADS * 2) cw = ELEM_THREADS * 2;
    chunks = (n + cw - 1) / cw;
    grid = dim3(chunks * rows, 1, 1);
}

hipError_t hp_launch_tensor(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 n, u32 P, const u64 *ct1,
                            const u64 *ct2, u64 *quad, hipStream_t stream) {
    if (kc == 0) return hipSuccess;
    u32 chunks, cw; dim3 grid;
    tensor_grid(n, P * kc, chunks, cw, grid);
    k_tensor<false><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, kc, n, chunks, cw, ct1, ct2, HpTensorRows{}, quad);
    return hipGetLastError();
}

hipError_t hp_launch_tensor_rows(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 n, u32 P, const HpTensorRows &rows, u64 *quad,
                                 hipStream_t stream) {
    if (kc == 0 || P == 0) return hipSuccess;
    if (P > HP_TENSOR_ROWS_MAX) return hipErrorInvalidValue;
    u32 chunks, cw; dim3 grid;
    tensor_grid(n, P * kc, chunks, cw, grid);
    k_tensor<true><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, kc, n, chunks, cw, nullptr, nullptr, rows, quad);
    return hipGetLastError();
}

// ---- key-switch inner product: rgsw.cpp:121-153 --------------------------------------------
// out[p][half][k][i] = montgomery_128( sum_j D[p][j][k][i] * key[j][half][k][i] ), 128-bit accumulators
// in registers, both halves from one pass over the digits.  Per (p, k, i): reads L digit words and 2L key
// words (the key is shared by the whole batch and stays in L2 / Infinity Cache), writes 2 words.
// One ciphertext (or an odd one out) per call: a LATENCY kernel -- hehub's one-call-per-ciphertext interface (ckks.h:270-313) puts a
// single key switch on the critical path of every call.  A workgroup covers 512 coefficients (two per lane, one pass), and the
// three 16-byte loads of FOUR digits are issued before their multiplications: the dependent rounds to memory drop from 4 x L to L / 4.
// MANY: every ciphertext of the launch has its OWN key (hp_dev_ckks_rotate_many: the rotations of one vector by different steps in
// the diagonal loop of src/circuits/linear_algebra.h:123-130); the key addresses travel as kernel arguments.  Nothing is shared
// between ciphertexts then, so this one-ciphertext-per-thread kernel is also the right one for a batch: 3L rows per (p, k).
#define KS1_CHUNK 512u
template <bool MANY>
__global__ void __launch_bounds__(ELEM_THREADS) k_ks_inner(const HpLimb *__restrict__ limbs, u32 L, u32 k_first, u32 P,
                                                          u32 key_Le, u32 n, u32 chunks, const u64 *__restrict__ digits,
                                                          const u64 *__restrict__ pt,
                                                          u32 pt_pstride, const u64 *__restrict__ key_one, HpKeyTable keys,
                                                          u64 *__restrict__ out) {
    const u32 Le = L + 1;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = k*P' ... decoded below
    // modulus-major numbering keeps one key column (2L limbs) hot per XCD slice
    const u32 k = k_first + row / P, p = row % P;
    const u64 *__restrict__ key = MANY ? keys.p[p] : key_one;
    const u64 q = limbs[k].q, mqinv = limbs[k].mqinv;
    const u32 i = chunk * KS1_CHUNK + threadIdx.x * 2;
    if (i >= n) return;
    const bool two = (i + 1 < n);
    // a key made for more moduli than the ciphertext has (extension): its special-prime column is the last one
    const u32 kcol = (k == L) ? key_Le - 1 : k;
    u64 a0l[2] = {0, 0}, a0h[2] = {0, 0}, a1l[2] = {0, 0}, a1h[2] = {0, 0};
    constexpr u32 G = 4;
    for (u32 j0 = 0; j0 < L; j0 += G) {
        u64 dv[G][2], k0[G][2], k1[G][2];
#pragma unroll
        for (u32 t = 0; t < G; t++) {
            const u32 j = j0 + t < L ? j0 + t : L - 1;   // (a slot past the last digit reloads it and is skipped below)
            const u64 *d = (j == k) ? pt + ((size_t)p * pt_pstride + j) * n : digits + (((size_t)p * L + j) * Le + k) * n;
            const u64 *g0 = key + (((size_t)j * 2 + 0) * key_Le + kcol) * n;
            const u64 *g1 = key + (((size_t)j * 2 + 1) * key_Le + kcol) * n;
            if (two) {
                // digits are read exactly once: non-temporal, so they do not evict the key column from L2
                typedef u64 __attribute__((ext_vector_type(2))) vv;
                const vv dvv = __builtin_nontemporal_load(reinterpret_cast<const vv *>(d + i));
                dv[t][0] = dvv.x; dv[t][1] = dvv.y;
                U2 w;
                w = *reinterpret_cast<const U2 *>(g0 + i); k0[t][0] = w.x; k0[t][1] = w.y;
                w = *reinterpret_cast<const U2 *>(g1 + i); k1[t][0] = w.x; k1[t][1] = w.y;
            } else {
                dv[t][0] = d[i]; k0[t][0] = g0[i]; k1[t][0] = g1[i]; dv[t][1] = k0[t][1] = k1[t][1] = 0;
            }
        }
#pragma unroll
        for (u32 t = 0; t < G; t++) {
            if (j0 + t >= L) break;
#pragma unroll
            for (int e = 0; e < 2; e++) {   // the sums in the reference's order j = 0 .. L-1 (rgsw.cpp:126-149; u128 addition is associative anyway)
                u64 lo, hi;
                hp_mul128(dv[t][e], k0[t][e], lo, hi);
                a0l[e] += lo; a0h[e] += hi + (a0l[e] < lo ? 1ull : 0ull);
                hp_mul128(dv[t][e], k1[t][e], lo, hi);
                a1l[e] += lo; a1h[e] += hi + (a1l[e] < lo ? 1ull : 0ull);
            }
        }
    }
    u64 *o0 = out + (((size_t)p * 2 + 0) * Le + k) * n;
    u64 *o1 = out + (((size_t)p * 2 + 1) * Le + k) * n;
    u64 r00 = hp_montgomery128_lazy(a0l[0], a0h[0], q, mqinv), r10 = hp_montgomery128_lazy(a1l[0], a1h[0], q, mqinv);
    if (two) {
        U2 v0{r00, hp_montgomery128_lazy(a0l[1], a0h[1], q, mqinv)};
        U2 v1{r10, hp_montgomery128_lazy(a1l[1], a1h[1], q, mqinv)};
        *reinterpret_cast<U2 *>(o0 + i) = v0;
        *reinterpret_cast<U2 *>(o1 + i) = v1;
    } else {
        o0[i] = r00; o1[i] = r10;
    }
}

// Same sums, PT ciphertexts per thread: the 2L key words of a (k, i) pair are loaded once and multiplied into PT
// ciphertexts' accumulators, so the key traffic through L2 / Infinity Cache (2L of the 3L+2 words per (p,k,i) above)
// drops by PT.  n is even for every supported ring (N >= 2) and chunks are even-sized: always two words per lane.
//
// Addressing is what bounded the first version of this kernel: a wave issued 218 SCALAR instructions per digit (64-bit row
// addresses for PT + 2 loads, each a multiply chain) next to 130 vector ones, and a SIMD issues at most one scalar instruction
// per turn -- VALUBusy 63 % with HBM at 57 %.  Now every stream is a buffer descriptor set up once per workgroup (PT digit
// rows, PT caller limbs for the diagonal, the key column), the lane offset is computed once per sweep, and the digit index
// moves ONE scalar offset per stream: ~10 scalar instructions per digit.
// The diagonal j == k (the caller's NTT-form limb, rgsw.cpp:99-101) comes first, then the L-1 (special prime: L) digit rows
// in a branch-free, hand double-buffered loop; u128 sums wrap, so the order of the terms does not matter.
typedef u32 __attribute__((ext_vector_type(4))) v4u;
typedef u32 __attribute__((ext_vector_type(2))) v2u;
constexpr int KS_NT = 2;   // buffer-load cache policy bit "nt": digits are read exactly once, keep them from evicting the key column

template <int PT> struct KsRow {
    U2 g0, g1;
    U2 d[PT];
};

HP_DEV __amdgpu_buffer_rsrc_t ks_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
HP_DEV U2 ks_u2(const v4u &v) { return U2{((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z}; }

template <int PT> HP_DEV void ks_mac(const KsRow<PT> &r, HpAcc (&acc)[PT][2][2]) {
    const u64 kw[2][2] = {{r.g0.x, r.g0.y}, {r.g1.x, r.g1.y}};
#pragma unroll
    for (int c = 0; c < PT; c++)
#pragma unroll
        for (int h = 0; h < 2; h++) hp_mac2(acc[c][h][0], r.d[c].x, kw[h][0], acc[c][h][1], r.d[c].y, kw[h][1]);
}

// PACK: 48 / 40 = the digit rows of this output modulus are in the HP_PACK48 / HP_PACK40 format (hp_device.h); 0 = plain words
template <int PT, int PACK>
HP_DEV void ks_sweep(const __amdgpu_buffer_rsrc_t (&rd)[PT], const __amdgpu_buffer_rsrc_t (&rp)[PT], __amdgpu_buffer_rsrc_t rk,
                     u32 i, u32 n, u32 L, u32 k, u32 d_stride, u32 k_stride, u32 k_half, u64 q, HpAcc (&acc)[PT][2][2]) {
    const u32 v16 = i << 3, v8 = i << 2, v4 = i << 1;   // lane byte offsets: plain words / low planes / high planes (48-bit rows)
    u64 ksum[2][2] = {{0, 0}, {0, 0}};                   // HP_PACK40: sum of the key words the offset rows were multiplied by
    const bool diag = k < L;
    const u32 T = diag ? L - 1 : L;                     // digit rows besides the diagonal
    auto load_key = [&](KsRow<PT> &r, u32 j) {
        const u32 so = __builtin_amdgcn_readfirstlane(j * k_stride);   // (wave-uniform: keeps the row offsets in SGPRs)
        r.g0 = ks_u2(__builtin_amdgcn_raw_buffer_load_b128(rk, v16, so, 0));
        r.g1 = ks_u2(__builtin_amdgcn_raw_buffer_load_b128(rk, v16, so + k_half, 0));
    };
    auto load_digit = [&](KsRow<PT> &r, u32 t) {
        const u32 j = t + ((diag && t >= k) ? 1u : 0u);
        load_key(r, j);
        const u32 so = __builtin_amdgcn_readfirstlane(j * d_stride);
#pragma unroll
        for (int c = 0; c < PT; c++) {
            if (PACK == 48) {
                const v2u lo = __builtin_amdgcn_raw_buffer_load_b64(rd[c], v8, so, KS_NT);
                const u32 hi = __builtin_amdgcn_raw_buffer_load_b32(rd[c], v4, so + (n << 2), KS_NT);
                r.d[c].x = lo.x | ((u64)(hi & 0xffffu) << 32);
                r.d[c].y = lo.y | ((u64)(hi >> 16) << 32);
            } else if (PACK == 40) {
                const v2u lo = __builtin_amdgcn_raw_buffer_load_b64(rd[c], v8, so, KS_NT);
                const u32 hi = (u32)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd[c], i, so + (n << 2), KS_NT);
                r.d[c].x = lo.x | ((u64)(hi & 0xffu) << 32);
                r.d[c].y = lo.y | ((u64)(hi >> 8) << 32);
            } else {
                r.d[c] = ks_u2(__builtin_amdgcn_raw_buffer_load_b128(rd[c], v16, so, KS_NT));
            }
        }
    };
    KsRow<PT> ra, rb;
    if (diag) {
        load_key(rb, k);
#pragma unroll
        for (int c = 0; c < PT; c++) rb.d[c] = ks_u2(__builtin_amdgcn_raw_buffer_load_b128(rp[c], v16, 0, KS_NT));
    }
    auto mac_digit = [&](const KsRow<PT> &r) {
        ks_mac<PT>(r, acc);
        if (PACK == 40) { ksum[0][0] += r.g0.x; ksum[0][1] += r.g0.y; ksum[1][0] += r.g1.x; ksum[1][1] += r.g1.y; }
    };
    if (T) load_digit(ra, 0);
    if (diag) ks_mac<PT>(rb, acc);
    if (!T) return;
    u32 t = 0;
    for (; t + 2 <= T; t += 2) {
        load_digit(rb, t + 1);
        mac_digit(ra);
        load_digit(ra, min(t + 2, T - 1));   // last: harmless re-read
        mac_digit(rb);
    }
    if (t < T) mac_digit(ra);
    if (PACK == 40) {
        const u64 qc = (q - 1) >> 1;
#pragma unroll
        for (int c = 0; c < PT; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) hp_mac2(acc[c][h][0], qc, ksum[h][0], acc[c][h][1], qc, ksum[h][1]);
    }
}

template <int PT, bool P40 = false>
__global__ void __launch_bounds__(ELEM_THREADS) k_ks_inner_blk(const HpLimb *__restrict__ limbs, u32 L, u32 k_first, u32 P,
                                                              u32 key_Le, u32 n, u32 chunks, const u64 *__restrict__ digits,
                                                              const u64 *__restrict__ pt, u32 pt_pstride,
                                                              const u64 *__restrict__ key, u64 *__restrict__ out, u32 pack_mask,
                                                              u32 pack40_mask) {
    const u32 Le = L + 1;
    const u32 PG = (P + PT - 1) / PT;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u32 k = k_first + row / PG, p0 = (row % PG) * PT;
    const u64 q = limbs[k].q, mqinv = limbs[k].mqinv;
    const u32 kcol = (k == L) ? key_Le - 1 : k;   // key made for more moduli (extension): special prime = its last column
    const bool packed = ((pack_mask >> k) & 1u) != 0, packed40 = P40 && ((pack40_mask >> k) & 1u) != 0;   // (P40: its own kernel, so that the level-B one keeps its registers)
    // descriptors: digit row (p, j = 0, k) -- the digit index adds j * Le * 8n bytes; the caller's limb (p, k); the key column
    // (j = 0, half 0, kcol) -- j adds 2 * key_Le * 8n bytes, the second half key_Le * 8n.  (32-bit offsets: L (L + 1) * 8n and
    // 2 L key_Le * 8n stay below 2^30 bytes at N = 32768 with the 32 limbs the engine allows.)
    __amdgpu_buffer_rsrc_t rd[PT], rp[PT];
#pragma unroll
    for (int c = 0; c < PT; c++) {
        const u32 p = min(p0 + c, P - 1);   // a ragged last group re-reads its last ciphertext and skips the store
        rd[c] = ks_rsrc(digits + ((size_t)p * L * Le + k) * n);
        rp[c] = ks_rsrc(pt + ((size_t)p * pt_pstride + min(k, L - 1)) * n);
    }
    const __amdgpu_buffer_rsrc_t rk = ks_rsrc(key + (size_t)kcol * n);
    const u32 d_stride = (Le * n) << 3, k_half = (key_Le * n) << 3, k_stride = k_half << 1;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x * 2; i < end; i += ELEM_THREADS * 2) {
        HpAcc acc[PT][2][2];   // [ciphertext][half][word]
#pragma unroll
        for (int c = 0; c < PT; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) { hp_acc_zero(acc[c][h][0]); hp_acc_zero(acc[c][h][1]); }
        if (P40 && packed40) ks_sweep<PT, 40>(rd, rp, rk, i, n, L, k, d_stride, k_stride, k_half, q, acc);
        else if (packed) ks_sweep<PT, 48>(rd, rp, rk, i, n, L, k, d_stride, k_stride, k_half, q, acc);
        else ks_sweep<PT, 0>(rd, rp, rk, i, n, L, k, d_stride, k_stride, k_half, q, acc);
#pragma unroll
        for (int c = 0; c < PT; c++) {
            const u32 p = p0 + c;
            if (p < P) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    u64 l0, h0, l1, h1;
                    hp_acc_value(acc[c][h][0], l0, h0);
                    hp_acc_value(acc[c][h][1], l1, h1);
                    U2 v{hp_montgomery128_lazy(l0, h0, q, mqinv), hp_montgomery128_lazy(l1, h1, q, mqinv)};
                    // (written once, read by the next kernel from HBM anyway: non-temporal, the key column keeps its place in L2)
                    typedef u64 __attribute__((ext_vector_type(2))) vv;
                    __builtin_nontemporal_store(vv{v.x, v.y}, reinterpret_cast<vv *>(out + (((size_t)p * 2 + h) * Le + k) * n + i));
                }
            }
        }
    }
}

hipError_t hp_launch_ks_inner(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 key_Le, u32 n, u32 P, const u64 *digits,
                              const u64 *pt, u32 pt_pstride, const u64 *key, u64 *out, u32 pack_mask, u32 pack40_mask,
                              hipStream_t stream) {
    if (kc == 0) return hipSuccess;
    if ((pack_mask | pack40_mask) && !(n >= 2 && P >= 2)) return hipErrorInvalidValue;   // the one-ciphertext kernel reads plain rows only
    // the blocked kernels address digit rows and key columns through buffer descriptors with 32-bit byte offsets
    // (j * (L+1) * 8n and j * 2 key_Le * 8n + key_Le * 8n, j < L): an offset past 2^31 would read zeros, not fault
    if ((u64)L * (L + 1) * 8u * n >= (1ull << 31) || 2ull * L * key_Le * 8u * n >= (1ull << 31)) return hipErrorInvalidValue;
    u32 chunks; dim3 grid;
    // ciphertexts per thread: four share every key word in registers (1 or 2 measured 1.5 % slower at the C3 shape)
    const int PT = (n >= 2 && P >= 4) ? 4 : (n >= 2 && P >= 2) ? 2 : 1;
    if (PT >= 4) {
        elem_grid(n, ((P + 3) / 4) * kc, chunks, grid);
        if (pack40_mask) k_ks_inner_blk<4, true><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride, key, out, pack_mask, pack40_mask);
        else k_ks_inner_blk<4><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride, key, out, pack_mask, 0u);
    } else if (PT >= 2) {
        elem_grid(n, ((P + 1) / 2) * kc, chunks, grid);
        if (pack40_mask) k_ks_inner_blk<2, true><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride, key, out, pack_mask, pack40_mask);
        else k_ks_inner_blk<2><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride, key, out, pack_mask, 0u);
    } else {
        chunks = (n + KS1_CHUNK - 1) / KS1_CHUNK;
        grid = dim3(chunks * P * kc, 1, 1);
        k_ks_inner<false><<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride, key, HpKeyTable{}, out);
    }
    return hipGetLastError();
}

// every ciphertext with its own key: plain digit rows, P <= HP_KEY_TABLE_MAX per launch
hipError_t hp_launch_ks_inner_many(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 key_Le, u32 n, u32 P, const u64 *digits,
                                   const u64 *pt, u32 pt_pstride, const HpKeyTable &keys, u64 *out, hipStream_t stream) {
    if (kc == 0 || P == 0) return hipSuccess;
    if (P > HP_KEY_TABLE_MAX) return hipErrorInvalidValue;
    const u32 chunks = (n + KS1_CHUNK - 1) / KS1_CHUNK;
    k_ks_inner<true><<<dim3(chunks * P * kc, 1, 1), ELEM_THREADS, 0, stream>>>(limbs, L, k_first, P, key_Le, n, chunks, digits, pt, pt_pstride,
                                                                               nullptr, keys, out);
    return hipGetLastError();
}

// ---- drop-last-prime helpers: rescaling.cpp:54-74 / mod_switch.cpp:52-76 --------------------
// rem[p2][k][i] = strict_barrett_{q_k}(c[i]) (+ q_k - r_k if c[i] >= q_last/2) (BGV: then * t)
__global__ void __launch_bounds__(ELEM_THREADS) k_drop_rem(const HpLimb *__restrict__ limbs, HpDropConsts dc, u32 Lm1,
                                                          u32 n, u32 chunks, const u64 *__restrict__ clast,
                                                          u64 *__restrict__ rem) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p2*Lm1 + k
    const u32 p2 = row / Lm1, k = row % Lm1;
    const u64 q = limbs[k].q, bc = limbs[k].barrett_c;
    const u64 bump = q - dc.r[k];
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        const u64 c = clast[(size_t)p2 * n + i];
        u64 v = hp_strict(hp_barrett_lazy(c, q, bc), q);
        if (c >= dc.half_q_last) v += bump;
        if (dc.bgv) v = hp_harvey_lazy(v, dc.t[k], dc.t_h[k], q);
        rem[(size_t)row * n + i] = v;
    }
}

hipError_t hp_launch_drop_rem(const HpLimb *limbs, const HpDropConsts &dc, u32 Lm1, u32 n, u32 P2, const u64 *clast,
                              u64 *rem, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P2 * Lm1, chunks, grid);
    k_drop_rem<<<grid, ELEM_THREADS, 0, stream>>>(limbs, dc, Lm1, n, chunks, clast, rem);
    return hipGetLastError();
}

// out = ((x - rem) * inv) [* (q_last mod t)] [+ addend]     (rns.cpp:89-118, :155-171, :58-87)
__global__ void __launch_bounds__(ELEM_THREADS) k_drop_fin(const HpLimb *__restrict__ limbs, HpDropConsts dc, u32 L,
                                                          u32 kc, u32 n, u32 chunks, const u64 *__restrict__ x,
                                                          const u64 *__restrict__ rem, const u64 *__restrict__ addend,
                                                          u32 add_poly_stride, u32 add_ct_stride, u32 add_mask,
                                                          u64 *__restrict__ out) {
    // kc limbs per polynomial are processed (all L-1, or a limb range whose first limb the pointers/constants
    // have been shifted to); x rows have stride L, out rows stride L-1, rem rows are compact [P2][kc]
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p2*kc + k
    const u32 p2 = row / kc, k = row % kc;
    const u64 q = limbs[k].q, two_q = limbs[k].two_q;
    const u64 *xs = x + ((size_t)p2 * L + k) * n;
    const u64 *as = (addend && ((add_mask >> (p2 & 1)) & 1u))
                        ? addend + ((size_t)(p2 >> 1) * add_ct_stride + (size_t)(p2 & 1) * add_poly_stride + k) * n : nullptr;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        u64 v = hp_sub_lazy(xs[i], rem[(size_t)row * n + i], two_q);
        v = hp_harvey_lazy(v, dc.inv[k], dc.inv_h[k], q);
        if (dc.bgv) v = hp_harvey_lazy(v, dc.qlt[k], dc.qlt_h[k], q);
        if (as) v = hp_add_lazy(v, as[i], two_q);
        out[((size_t)p2 * (L - 1) + k) * n + i] = v;
    }
}

hipError_t hp_launch_drop_fin(const HpLimb *limbs, const HpDropConsts &dc, u32 L, u32 kc, u32 n, u32 P2, const u64 *x,
                              const u64 *rem, const u64 *addend, u32 add_poly_stride, u32 add_ct_stride, u32 add_mask,
                              u64 *out, hipStream_t stream) {
    if (kc == 0) return hipSuccess;
    u32 chunks; dim3 grid;
    elem_grid(n, P2 * kc, chunks, grid);
    k_drop_fin<<<grid, ELEM_THREADS, 0, stream>>>(limbs, dc, L, kc, n, chunks, x, rem, addend, add_poly_stride,
                                                  add_ct_stride, add_mask, out);
    return hipGetLastError();
}

// ---- either side of the path: encrypt / decrypt cores and RNS base transforms ------
// Simple one-word-per-lane streaming kernels: these run once per ciphertext, not once per multiplication.

// sampling.cpp:77-83: ex[p][k][i] = q_k + (u64)e[p][i], minus q_k if that reached q_k
__global__ void __launch_bounds__(ELEM_THREADS) k_lift_noise(const HpLimb *__restrict__ limbs, u32 L, u32 n, u32 chunks,
                                                            const long long *__restrict__ noise, u64 *__restrict__ out,
                                                            u32 out_pstride) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*L + k
    const u32 p = row / L, k = row % L;
    const u64 q = limbs[k].q;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        u64 v = q + (u64)noise[(size_t)p * n + i];
        out[((size_t)p * out_pstride + k) * n + i] = v - ((v >= q) ? q : 0);
    }
}

// rlwe.cpp:52 and :70: c0 = (ex - c1*sk) + NTT(pt); ct[p] = (c0, c1).  ex is read from ct[p][0] (in place).
__global__ void __launch_bounds__(ELEM_THREADS) k_enc_fin(const HpLimb *__restrict__ limbs, u32 L, u32 n, u32 chunks,
                                                         const u64 *__restrict__ c1, const u64 *__restrict__ sk,
                                                         const u64 *__restrict__ ptn, u64 *__restrict__ ct) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*L + k
    const u32 p = row / L, k = row % L;
    const HpLimb m = limbs[k];
    u64 *c0 = ct + ((size_t)p * 2 * L + k) * n, *o1 = c0 + (size_t)L * n;
    const u64 *a = c1 + (size_t)row * n, *s = sk + (size_t)k * n, *t = ptn + (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        const u64 av = a[i];
        u64 v = hp_sub_lazy(c0[i], hp_mul_hybrid_lazy(av, s[i], m), m.two_q);
        c0[i] = hp_add_lazy(v, t[i], m.two_q);
        o1[i] = av;
    }
}

// rlwe.cpp:76: c0 + c1*sk (the INTT and the strict reduction follow as a transform launch)
__global__ void __launch_bounds__(ELEM_THREADS) k_dec_fma(const HpLimb *__restrict__ limbs, u32 L, u32 n, u32 chunks,
                                                         const u64 *__restrict__ ct, const u64 *__restrict__ sk,
                                                         u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*L + k
    const u32 p = row / L, k = row % L;
    const HpLimb m = limbs[k];
    const u64 *c0 = ct + ((size_t)p * 2 * L + k) * n, *c1 = c0 + (size_t)L * n, *s = sk + (size_t)k * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS)
        out[(size_t)row * n + i] = hp_add_lazy(c0[i], hp_mul_hybrid_lazy(c1[i], s[i], m), m.two_q);
}

// rns_transform.cpp:113 + :11-37: strict(x) mod old -> centred lift into every new modulus (+ lazy Barrett when q < old)
__global__ void __launch_bounds__(ELEM_THREADS) k_base_from_single(const HpLimb *__restrict__ limbs, u64 old_q, u32 L, u32 n,
                                                                  u32 chunks, const u64 *__restrict__ in,
                                                                  u64 *__restrict__ out) {
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*L + k
    const u32 p = row / L, k = row % L;
    const u64 q = limbs[k].q, bc = limbs[k].barrett_c;
    const u64 half = old_q / 2, multiple = (old_q / q + 1) * q;
    const bool reduce = q < old_q;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        const u64 x = hp_strict(in[(size_t)p * n + i], old_q);
        u64 v = (x < half) ? x : multiple - old_q + x;
        if (reduce) v = hp_barrett_lazy(v, q, bc);
        out[(size_t)row * n + i] = v;
    }
}

// rns_transform.cpp:113 + :39-84 (small-coefficient branch): consistency check over the limbs (any violation sets
// not_small[p]) and the centred lift of limb 0 into the new modulus, strictly Barrett-reduced
__global__ void __launch_bounds__(ELEM_THREADS) k_base_to_single(const HpLimb *__restrict__ limbs, u32 L, u32 n, u32 chunks,
                                                                u64 new_q, u64 new_bc, const u64 *__restrict__ in,
                                                                u64 *__restrict__ out, u32 *__restrict__ not_small) {
    const u32 p = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u64 q0 = limbs[0].q, half = q0 / 2, multiple = (q0 / new_q + 1) * new_q;
    const u64 *x = in + (size_t)p * L * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    bool bad = false;
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        const u64 x0 = hp_strict(x[i], q0);
        for (u32 k = 1; k < L; k++) {
            const u64 qk = limbs[k].q, xk = hp_strict(x[(size_t)k * n + i], qk);
            bad |= (x0 < half) ? (xk != x0) : (qk - xk != q0 - x0);
        }
        const u64 v = (x0 < half) ? x0 : multiple - q0 + x0;
        out[(size_t)p * n + i] = hp_strict(hp_barrett_lazy(v, new_q, new_bc), new_q);
    }
    if (bad) atomicOr(not_small + p, 1u);
}

hipError_t hp_launch_lift_noise(const HpLimb *limbs, u32 L, u32 n, u32 P, const long long *noise, u64 *out, u32 out_pstride,
                                hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P * L, chunks, grid);
    k_lift_noise<<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, noise, out, out_pstride);
    return hipGetLastError();
}
hipError_t hp_launch_enc_fin(const HpLimb *limbs, u32 L, u32 n, u32 P, const u64 *c1, const u64 *sk, const u64 *ptn, u64 *ct,
                             hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P * L, chunks, grid);
    k_enc_fin<<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, c1, sk, ptn, ct);
    return hipGetLastError();
}
hipError_t hp_launch_dec_fma(const HpLimb *limbs, u32 L, u32 n, u32 P, const u64 *ct, const u64 *sk, u64 *out,
                             hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P * L, chunks, grid);
    k_dec_fma<<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, ct, sk, out);
    return hipGetLastError();
}
hipError_t hp_launch_base_from_single(const HpLimb *limbs, u64 old_q, u32 L, u32 n, u32 P, const u64 *in, u64 *out,
                                      hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P * L, chunks, grid);
    k_base_from_single<<<grid, ELEM_THREADS, 0, stream>>>(limbs, old_q, L, n, chunks, in, out);
    return hipGetLastError();
}
hipError_t hp_launch_base_to_single(const HpLimb *limbs, u32 L, u32 n, u32 P, u64 new_q, const u64 *in, u64 *out,
                                    u32 *not_small, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P, chunks, grid);
    hipError_t e = hipMemsetAsync(not_small, 0, (size_t)P * sizeof(u32), stream);
    if (e != hipSuccess) return e;
    k_base_to_single<<<grid, ELEM_THREADS, 0, stream>>>(limbs, L, n, chunks, new_q, (~(u64)0) / new_q, in, out, not_small);
    return hipGetLastError();
}

// rns_transform.cpp:86-104 on the device, without big integers: mixed-radix (Garner) digits v_i of the CRT value
// x = v_0 + v_1 q_0 + v_2 q_0 q_1 + ... (0 <= v_i < q_i) are computed with word arithmetic, x < floor(Q/2) is a
// lexicographic comparison with the digits of floor(Q/2), and x mod t is sum v_i (q_0...q_{i-1} mod t).  The result is the
// reference's: x mod t below the half, t - ((Q - x) mod t) from the half on (which is t itself, not 0, when t | Q - x).
// Only polynomials flagged not_small are touched; the others keep the small-coefficient result.
__global__ void __launch_bounds__(ELEM_THREADS) k_base_to_single_crt(const HpLimb *__restrict__ limbs, const HpCrtConsts *__restrict__ cc,
                                                                    u32 L, u32 n, u32 chunks, const u64 *__restrict__ in,
                                                                    u64 *__restrict__ out, u32 out_pstride,
                                                                    const u32 *__restrict__ not_small) {
    const u32 p = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    if (not_small && !not_small[p]) return;   // NULL: every polynomial takes the CRT composition
    const u64 t = cc->t;
    const u64 *x = in + (size_t)p * L * n;
    const u32 end = min(n, (chunk + 1) * ELEM_CHUNK);
    for (u32 i = chunk * ELEM_CHUNK + threadIdx.x; i < end; i += ELEM_THREADS) {
        u64 v[HP_CRT_MAX_LIMBS];
        for (u32 a = 0; a < L; a++) {
            const u64 qa = limbs[a].q, bc = limbs[a].barrett_c;
            u64 u = hp_strict(x[(size_t)a * n + i], qa);
            for (u32 b = 0; b < a; b++) {
                const u64 vb = hp_strict(hp_barrett_lazy(v[b], qa, bc), qa);          // v_b mod q_a
                const u64 d = u + qa - vb;                                            // in (0, 2 q_a)
                u = hp_strict(hp_harvey_lazy(d, cc->inv[b][a], cc->inv_h[b][a], qa), qa);   // (u - v_b) / q_b mod q_a
            }
            v[a] = u;
        }
        bool below = false, decided = false;   // x < floor(Q/2): compare digits from the most significant one down
        for (int a = (int)L - 1; a >= 0 && !decided; a--) {
            if (v[a] != cc->half[a]) { below = v[a] < cc->half[a]; decided = true; }
        }
        u64 r = 0;   // x mod t
        for (u32 a = 0; a < L; a++) {
            r += hp_strict(hp_harvey_lazy(v[a], cc->pref[a], cc->pref_h[a], t), t);
            r -= (r >= t) ? t : 0;
        }
        u64 res;
        if (below) {
            res = r;
        } else {
            u64 abs = cc->q_mod_t + t - r;   // (Q - x) mod t
            abs -= (abs >= t) ? t : 0;
            res = t - abs;
        }
        out[(size_t)p * out_pstride * n + i] = res;
    }
}

hipError_t hp_launch_base_to_single_crt(const HpLimb *limbs, const HpCrtConsts *cc, u32 L, u32 n, u32 P, const u64 *in, u64 *out,
                                        u32 out_pstride, const u32 *not_small, hipStream_t stream) {
    u32 chunks; dim3 grid;
    elem_grid(n, P, chunks, grid);
    k_base_to_single_crt<<<grid, ELEM_THREADS, 0, stream>>>(limbs, cc, L, n, chunks, in, out, out_pstride, not_small);
    return hipGetLastError();
}
