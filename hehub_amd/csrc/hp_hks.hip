// hp_hks.hip -- kernels of the hybrid key switch (EXTENSION, not part of hehub: see include/hehub_amd.h).
//
// hehub switches keys with one digit per RNS limb and one special prime: L digits, L*(L+1) digit transforms per
// switch (rgsw.cpp:57-156).  The hybrid variant groups alpha consecutive limbs into one digit (dnum = ceil(L/alpha)
// digits) and uses k special primes P = p_0...p_{k-1} >= the largest digit: dnum*(L+k) - L transforms instead of L*L.
// It needs keys in its own format (one row per digit), so it can never be bit-compatible with hehub's keys; it is
// pinned by an exact integer model and by decryption (tests/test_hks.py).
//
//   ModUp    digit d = limbs [d*alpha, ...): the EXACT integer x_d in [0, Q_d) behind its residues (mixed-radix / Garner
//            digits, word arithmetic) reduced into every other modulus of q_0..q_{L-1}, p_0..p_{k-1}
//   inner    out[half][m] = montgomery( sum_d D[d][m] * key[d][half][m] ), D[d][m] = NTT_m(lift) or the input limb itself
//            when m belongs to digit d -- the same 128-bit accumulation as hehub's inner product
//   ModDown  the P-part of the result, centred exactly (hp_elem.hip: k_base_to_single_crt), is subtracted and the rest
//            multiplied by P^-1:  out = (x - NTT(rem)) * P^-1 [+ addend]
#include "hp_kernels.h"
#include <cstdlib>

struct alignas(16) U2 {
    u64 x, y;
};

#define HKS_THREADS 256
#define HKS_CHUNK 2048u

static inline void hks_grid(u32 n, u32 rows, u32 &chunks, dim3 &grid) {
    chunks = (n + HKS_CHUNK - 1) / HKS_CHUNK;
    grid = dim3(chunks * rows, 1, 1);
}

// lifted[p][d][m][i] = x_d mod modulus_m for every modulus m outside digit d (slots inside the digit are not written).
// Two coefficients per lane and iteration: 16-byte accesses (n is even for every supported ring).
template <int ALPHA>
__global__ void __launch_bounds__(HKS_THREADS) k_hks_modup(const HpLimb *__restrict__ limbs, const HpHksConsts *__restrict__ hc,
                                                          u32 n, u32 chunks, const u64 *__restrict__ coef,
                                                          u64 *__restrict__ lifted) {
    const u32 L = hc->L, E = hc->E, nd = hc->nd;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p*nd + d
    const u32 p = row / nd, d = row % nd;
    const u32 first = d * hc->alpha, cnt = min(hc->alpha, L - first);   // limbs of this digit
    const u64 *src = coef + ((size_t)p * L + first) * n;
    u64 *dst = lifted + (size_t)row * E * n;
    const u32 end = min(n, (chunk + 1) * HKS_CHUNK);
    for (u32 i = chunk * HKS_CHUNK + threadIdx.x * 2; i < end; i += HKS_THREADS * 2) {
        u64 v[ALPHA][2];
#pragma unroll
        for (int a = 0; a < ALPHA; a++) {
            if ((u32)a < cnt) {
                const u64 qa = limbs[first + a].q, bc = limbs[first + a].barrett_c;
                const U2 in = *reinterpret_cast<const U2 *>(src + (size_t)a * n + i);   // strict residues
                u64 u[2] = {in.x, in.y};
#pragma unroll
                for (int b = 0; b < a; b++) {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const u64 vb = hp_strict(hp_barrett_lazy(v[b][e], qa, bc), qa);
                        u[e] = hp_strict(hp_harvey_lazy(u[e] + qa - vb, hc->inv[d][b][a], hc->inv_h[d][b][a], qa), qa);
                    }
                }
                v[a][0] = u[0];
                v[a][1] = u[1];
            } else {
                v[a][0] = v[a][1] = 0;
            }
        }
        for (u32 m = 0; m < E; m++) {
            if (m >= first && m < first + cnt) continue;
            const u64 qm = limbs[m].q;
            u64 r[2] = {0, 0};
#pragma unroll
            for (int a = 0; a < ALPHA; a++) {
                if ((u32)a < cnt) {
                    const u64 w = hc->pref[d][m][a], wh = hc->pref_h[d][m][a];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        r[e] += hp_strict(hp_harvey_lazy(v[a][e], w, wh, qm), qm);
                        r[e] -= (r[e] >= qm) ? qm : 0;
                    }
                }
            }
            U2 o{r[0], r[1]};
            *reinterpret_cast<U2 *>(dst + (size_t)m * n + i) = o;
        }
    }
}

hipError_t hp_launch_hks_modup(const HpLimb *limbs, const HpHksConsts *hc, u32 alpha, u32 nd, u32 n, u32 P, const u64 *coef,
                               u64 *lifted, hipStream_t stream) {
    u32 chunks; dim3 grid;
    hks_grid(n, P * nd, chunks, grid);
    switch (alpha) {
    case 1: k_hks_modup<1><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 2: k_hks_modup<2><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 3: k_hks_modup<3><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 4: k_hks_modup<4><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 5: k_hks_modup<5><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 6: k_hks_modup<6><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 7: k_hks_modup<7><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    case 8: k_hks_modup<8><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, coef, lifted); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// out[p][half][m][i] = montgomery_128( sum_d D[p][d][m][i] * key[d][half][m][i] ); PT ciphertexts share each key word
template <int PT>
__global__ void __launch_bounds__(HKS_THREADS) k_hks_inner(const HpLimb *__restrict__ limbs, u32 L, u32 E, u32 nd, u32 alpha, u32 P,
                                                          u32 n, u32 chunks, const u64 *__restrict__ lifted,
                                                          const u64 *__restrict__ pt, u32 pt_pstride,
                                                          const u64 *__restrict__ key, u64 *__restrict__ out) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    const u32 PG = (P + PT - 1) / PT;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u32 m = row / PG, p0 = (row % PG) * PT;
    const u64 q = limbs[m].q, mqinv = limbs[m].mqinv;
    const u32 own = (m < L) ? m / alpha : nd;   // the digit this modulus belongs to (none for the special primes)
    const u32 end = min(n, (chunk + 1) * HKS_CHUNK);
    for (u32 i = chunk * HKS_CHUNK + threadIdx.x * 2; i < end; i += HKS_THREADS * 2) {
        HpAcc acc[PT][2][2];   // carry-save columns (hp_device.h)
#pragma unroll
        for (int c = 0; c < PT; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) { hp_acc_zero(acc[c][h][0]); hp_acc_zero(acc[c][h][1]); }
        for (u32 d = 0; d < nd; d++) {
            const U2 g0 = *reinterpret_cast<const U2 *>(key + (((size_t)d * 2 + 0) * E + m) * n + i);
            const U2 g1 = *reinterpret_cast<const U2 *>(key + (((size_t)d * 2 + 1) * E + m) * n + i);
            const u64 kw[2][2] = {{g0.x, g0.y}, {g1.x, g1.y}};
#pragma unroll
            for (int c = 0; c < PT; c++) {
                const u32 p = min(p0 + c, P - 1);
                const u64 *src = (d == own) ? pt + ((size_t)p * pt_pstride + m) * n : lifted + (((size_t)p * nd + d) * E + m) * n;
                const vv t = __builtin_nontemporal_load(reinterpret_cast<const vv *>(src + i));
#pragma unroll
                for (int h = 0; h < 2; h++) hp_mac2(acc[c][h][0], t.x, kw[h][0], acc[c][h][1], t.y, kw[h][1]);
            }
        }
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
                    __builtin_nontemporal_store(vv{v.x, v.y}, reinterpret_cast<vv *>(out + (((size_t)p * 2 + h) * E + m) * n + i));   // written once, read from HBM by the next kernel
                }
            }
        }
    }
}

hipError_t hp_launch_hks_inner(const HpLimb *limbs, u32 L, u32 E, u32 nd, u32 alpha, u32 n, u32 P, const u64 *lifted, const u64 *pt,
                               u32 pt_pstride, const u64 *key, u64 *out, hipStream_t stream) {
    u32 chunks; dim3 grid;
    // two ciphertexts per thread (four: the column accumulators halve the occupancy, measured -5 %)
    if (P >= 2) {
        hks_grid(n, ((P + 1) / 2) * E, chunks, grid);
        k_hks_inner<2><<<grid, HKS_THREADS, 0, stream>>>(limbs, L, E, nd, alpha, P, n, chunks, lifted, pt, pt_pstride, key, out);
    } else {
        hks_grid(n, P * E, chunks, grid);
        k_hks_inner<1><<<grid, HKS_THREADS, 0, stream>>>(limbs, L, E, nd, alpha, P, n, chunks, lifted, pt, pt_pstride, key, out);
    }
    return hipGetLastError();
}

// ModDown conversion: Garner digits of the special-prime part once per coefficient, then its exact centred value
// (x below floor(P/2), x - P from there on; an exact multiple of q_i above the half comes out as q_i, a representative
// of 0) in every ciphertext modulus
template <int K>
__global__ void __launch_bounds__(HKS_THREADS) k_hks_moddown(const HpLimb *__restrict__ limbs, const HpHksConsts *__restrict__ hc,
                                                            u32 n, u32 chunks, const u64 *__restrict__ yp, u64 *__restrict__ rem) {
    const u32 L = hc->L;
    const u32 p2 = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const u64 *src = yp + (size_t)p2 * K * n;
    u64 *dst = rem + (size_t)p2 * L * n;
    const u32 end = min(n, (chunk + 1) * HKS_CHUNK);
    for (u32 i = chunk * HKS_CHUNK + threadIdx.x; i < end; i += HKS_THREADS) {
        u64 v[K];
#pragma unroll
        for (int a = 0; a < K; a++) {
            const u64 pa = limbs[L + a].q, bc = limbs[L + a].barrett_c;
            u64 u = src[(size_t)a * n + i];
#pragma unroll
            for (int b = 0; b < a; b++) {
                const u64 vb = hp_strict(hp_barrett_lazy(v[b], pa, bc), pa);
                u = hp_strict(hp_harvey_lazy(u + pa - vb, hc->pg_inv[b][a], hc->pg_inv_h[b][a], pa), pa);
            }
            v[a] = u;
        }
        bool below = false, decided = false;
#pragma unroll
        for (int a = K - 1; a >= 0; a--) {
            if (!decided && v[a] != hc->p_half[a]) { below = v[a] < hc->p_half[a]; decided = true; }
        }
        for (u32 m = 0; m < L; m++) {
            const u64 qm = limbs[m].q;
            u64 r = 0;
#pragma unroll
            for (int a = 0; a < K; a++) {
                r += hp_strict(hp_harvey_lazy(v[a], hc->p_pref[m][a], hc->p_pref_h[m][a], qm), qm);
                r -= (r >= qm) ? qm : 0;
            }
            if (!below) {
                u64 abs = hc->p_mod_q[m] + qm - r;
                abs -= (abs >= qm) ? qm : 0;
                r = qm - abs;
            }
            dst[(size_t)m * n + i] = r;
        }
    }
}

hipError_t hp_launch_hks_moddown(const HpLimb *limbs, const HpHksConsts *hc, u32 k, u32 n, u32 P2, const u64 *yp, u64 *rem,
                                 hipStream_t stream) {
    u32 chunks; dim3 grid;
    hks_grid(n, P2, chunks, grid);
    switch (k) {
    case 1: k_hks_moddown<1><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 2: k_hks_moddown<2><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 3: k_hks_moddown<3><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 4: k_hks_moddown<4><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 5: k_hks_moddown<5><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 6: k_hks_moddown<6><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 7: k_hks_moddown<7><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    case 8: k_hks_moddown<8><<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, yp, rem); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// merged ModDown + rescale: the remainder of the division by q_{L-1} joins the ModDown remainder before the transform.
// In the coefficient domain, limb i < L-1:  rem_i <- rem_i + (P mod q_i) * centre_{q_i}(c), c the strict coefficient of the
// relinearised limb L-1 modulo q_last (centred like rescaling.cpp:54-69: c >= q_last/2 means c - q_last)
__global__ void __launch_bounds__(HKS_THREADS) k_hks_combine(const HpLimb *__restrict__ limbs, const HpHksConsts *__restrict__ hc,
                                                            u32 n, u32 chunks, const u64 *__restrict__ clast,
                                                            u64 *__restrict__ rem) {
    const u32 L = hc->L, Lm1 = L - 1;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p2*(L-1) + i
    const u32 p2 = row / Lm1, k = row % Lm1;
    const u64 q = limbs[k].q, bc = limbs[k].barrett_c, two_q = limbs[k].two_q;
    const u64 q_last = limbs[Lm1].q, half = q_last / 2;
    const u64 bump = q - hp_strict(hp_barrett_lazy(q_last, q, bc), q);   // q_i - (q_last mod q_i)
    const u64 pm = hc->p_mod_q[k], pmh = hc->p_mod_q_h[k];
    const u64 *c = clast + (size_t)p2 * n;
    u64 *r = rem + ((size_t)p2 * L + k) * n;
    const u32 end = min(n, (chunk + 1) * HKS_CHUNK);
    for (u32 i = chunk * HKS_CHUNK + threadIdx.x; i < end; i += HKS_THREADS) {
        const u64 cv = c[i];
        u64 v = hp_strict(hp_barrett_lazy(cv, q, bc), q);
        if (cv >= half) v += bump;                         // < 2 q_i
        v = hp_harvey_lazy(v, pm, pmh, q);                 // * (P mod q_i), lazy
        r[i] = hp_add_lazy(r[i], v, two_q);
    }
}

hipError_t hp_launch_hks_combine(const HpLimb *limbs, const HpHksConsts *hc, u32 L, u32 n, u32 P2, const u64 *clast, u64 *rem,
                                 hipStream_t stream) {
    if (L < 2) return hipSuccess;
    u32 chunks; dim3 grid;
    hks_grid(n, P2 * (L - 1), chunks, grid);
    k_hks_combine<<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, clast, rem);
    return hipGetLastError();
}

// ModDown epilogue: out[p2][i] = ((x[p2][i] - rem[p2][i]) * P^-1 mod q_i) [+ addend]; x rows have E limbs, rem / out rows L
__global__ void __launch_bounds__(HKS_THREADS) k_hks_down_fin(const HpLimb *__restrict__ limbs, const HpHksConsts *__restrict__ hc,
                                                             u32 n, u32 chunks, const u64 *__restrict__ x,
                                                             const u64 *__restrict__ rem, const u64 *__restrict__ addend,
                                                             u32 add_poly_stride, u32 add_ct_stride, u32 add_mask,
                                                             u64 *__restrict__ out) {
    const u32 L = hc->L, E = hc->E;
    const u32 row = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // row = p2*L + i
    const u32 p2 = row / L, k = row % L;
    const u64 q = limbs[k].q, two_q = limbs[k].two_q;
    const u64 *xs = x + ((size_t)p2 * E + k) * n;
    const u64 *rs = rem + (size_t)row * n;
    const u64 *as = (addend && ((add_mask >> (p2 & 1)) & 1u))
                        ? addend + ((size_t)(p2 >> 1) * add_ct_stride + (size_t)(p2 & 1) * add_poly_stride + k) * n : nullptr;
    u64 *os = out + (size_t)row * n;
    const u32 end = min(n, (chunk + 1) * HKS_CHUNK);
    for (u32 i = chunk * HKS_CHUNK + threadIdx.x; i < end; i += HKS_THREADS) {
        u64 v = hp_sub_lazy(xs[i], rs[i], two_q);
        v = hp_harvey_lazy(v, hc->pinv[k], hc->pinv_h[k], q);
        if (as) v = hp_add_lazy(v, as[i], two_q);
        os[i] = v;
    }
}

hipError_t hp_launch_hks_down_fin(const HpLimb *limbs, const HpHksConsts *hc, u32 L, u32 n, u32 P2, const u64 *x, const u64 *rem,
                                  const u64 *addend, u32 add_poly_stride, u32 add_ct_stride, u32 add_mask, u64 *out,
                                  hipStream_t stream) {
    u32 chunks; dim3 grid;
    hks_grid(n, P2 * L, chunks, grid);
    k_hks_down_fin<<<grid, HKS_THREADS, 0, stream>>>(limbs, hc, n, chunks, x, rem, addend, add_poly_stride, add_ct_stride, add_mask, out);
    return hipGetLastError();
}
