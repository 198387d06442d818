// Level-A candidate butterfly on gfx950: the residue x*w mod q through error-free FP64 products instead of the exact Harvey
// quotient (hp_butterfly2_nq: 16 VALU instructions, 61 cycles per wave-butterfly).  Valid for q < 2^50, |x| < 2^52.
//   h = RN(x w), l = fma(x, w, -h)           x w = h + l exactly
//   k = rint(x u), u = RN(w / q)             |k - x w / q| <= 1/2 + |x| 2^-52
//   t = fma(-k, q, h) + l                    = x w - k q exactly, |t| <= q (1/2 + |x| 2^-52)
//   hi = lo - t, lo = lo + t                 8 FP64 instructions
// Same harness as ubench_bfly.hip (32 coefficients per thread, 5-stage passes in a loop) + an exactness check of t against
// 128-bit integer arithmetic on the device.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../hehub_amd/csrc -o ubench_bfly_f64 ubench_bfly_f64.hip
#include "hp_device.h"
#include <cstdio>
#include <cmath>
#include <vector>

#define FDEV __device__ __forceinline__

template <int RND> FDEV double f_modmul(double x, double w, double u, double q) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    double k;
    if (RND == 0) k = __builtin_rint(x * u);
    else { const double M = 6755399441055744.0; k = __builtin_fma(x, u, M) - M; }
    const double r = __builtin_fma(-k, q, h);
    return r + l;
}
template <int RND> FDEV void f_bfly(double &lo, double &hi, double w, double u, double q) {
    const double t = f_modmul<RND>(hi, w, u, q);
    hi = lo - t;
    lo = lo + t;
}
FDEV double f_reduce(double x, double qinv, double q) { return __builtin_fma(-__builtin_rint(x * qinv), q, x); }

// VARIANT 0: integer asm dual butterfly (today's), 1: f64 rint, 2: f64 magic rounding, 3: f64 rint + one range reduction per pass
template <int VARIANT> __global__ void __launch_bounds__(256, 4) kern(u64 *out, const u64 *tw, u64 q, int iters) {
    if constexpr (VARIANT == 0) {
        u64 x[32];
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = out[threadIdx.x + 256 * r];
        const u64 two_q = 2 * q, nq = 0 - q;
        u64 w = tw[threadIdx.x & 7], wh = tw[8 + (threadIdx.x & 7)];
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int b = 4; b >= 0; --b) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const int ra = ((i >> b) << (b + 1)) | (i & ((1 << b) - 1));
                    const int rb = (((i + 1) >> b) << (b + 1)) | ((i + 1) & ((1 << b) - 1));
                    hp_butterfly2_nq(x[ra], x[ra | (1 << b)], x[rb], x[rb | (1 << b)], w, wh, w, wh, two_q, (u32)nq, (u32)(nq >> 32));
                }
                w += 2; wh += 3;
            }
        }
#pragma unroll
        for (int r = 0; r < 32; r++) out[threadIdx.x + 256 * r] = x[r];
    } else {
        double x[32];
        const double qd = (double)q, qinv = 1.0 / qd;
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = (double)(out[threadIdx.x + 256 * r] % q);
        double w = (double)(tw[threadIdx.x & 7] % q), u = w / qd;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int b = 4; b >= 0; --b) {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    if (r & (1 << b)) continue;
                    f_bfly<VARIANT == 2 ? 1 : 0>(x[r], x[r | (1 << b)], w, u, qd);
                }
                w += 2.0; u += 1e-13;   // keep the compiler from hoisting anything
            }
            if (VARIANT == 3) {
#pragma unroll
                for (int r = 0; r < 32; ++r) x[r] = f_reduce(x[r], qinv, qd);
            } else {
                // (keep the values bounded so that the timing loop does not run into infinities: not counted as butterfly work,
                // one instruction per coefficient per pass)
#pragma unroll
                for (int r = 0; r < 32; ++r) x[r] *= 0.03125;
            }
        }
#pragma unroll
        for (int r = 0; r < 32; r++) out[threadIdx.x + 256 * r] = (u64)(long long)x[r];
    }
}

template <int VARIANT> void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 200;
    u64 *d, *tw;
    (void)hipMalloc(&d, 256 * 32 * 8); (void)hipMalloc(&tw, 16 * 8);
    (void)hipMemset(d, 1, 256 * 32 * 8); (void)hipMemset(tw, 3, 16 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<VARIANT><<<blocks, 256>>>(d, tw, 1099510054913ull, iters);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<VARIANT><<<blocks, 256>>>(d, tw, 1099510054913ull, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double bf_per_simd = (double)iters * 80 * blocks_per_cu;
    printf("%-44s waves/SIMD=%d : %8.3f ms  -> %.1f cycles per wave-butterfly per SIMD (2.1 GHz)\n", name, blocks_per_cu, best,
           best * 1e-3 * 2.1e9 / bf_per_simd);
    (void)hipFree(d); (void)hipFree(tw);
}

// ---- exactness: t == x w - k q with |t| <= q (1/2 + |x| 2^-52) and t = x w (mod q), against 128-bit integers ----
__device__ u64 splitmix(u64 &s) {
    u64 z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void check(u64 q, int xbits, int per_thread, unsigned long long *bad, double *worst) {
    u64 s = 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x + 1) + q;
    const double qd = (double)q;
    double wmax = 0;
    for (int i = 0; i < per_thread; ++i) {
        const u64 wi = splitmix(s) % q;
        u64 xm = splitmix(s) & ((1ull << xbits) - 1);
        if ((i & 7) == 0) xm = (1ull << xbits) - 1 - (xm & 15);       // edge: largest magnitudes
        if ((i & 15) == 1) xm = xm & 15;                               // edge: tiny
        const bool neg = splitmix(s) & 1;
        const double x = neg ? -(double)xm : (double)xm, w = (double)wi, u = w / qd;
        const double t0 = f_modmul<0>(x, w, u, qd), t1 = f_modmul<1>(x, w, u, qd);
        // reference residue
        const unsigned __int128 p = (unsigned __int128)xm * wi;
        u64 ref = (u64)(p % q);
        if (neg && ref) ref = q - ref;
        for (int v = 0; v < 2; ++v) {
            const double t = v ? t1 : t0;
            if (v == 1 && xbits > 50) continue;                         // magic rounding needs |x u| < 2^51
            long long ti = (long long)t;
            bool ok = ((double)ti == t);
            long long m = ti % (long long)q; if (m < 0) m += q;
            ok = ok && ((u64)m == ref);
            const double bound = qd * (0.5 + fabs(x) * 0x1p-52) + 1.0;
            ok = ok && fabs(t) <= bound;
            if (!ok) atomicAdd(bad, 1ull);
            const double rel = fabs(t) / qd;
            if (rel > wmax) wmax = rel;
        }
    }
    if (wmax > 0.5) { /* racy max is good enough for a report */ if (wmax > *worst) *worst = wmax; }
}

int main() {
    const u64 qs[] = {1099510054913ull, 1125899904679937ull, 1125899903827969ull, 65537ull, 576460752272228353ull >> 10 | 1};
    for (u64 q : qs) for (int xbits : {40, 50, 52}) {
        unsigned long long *bad; double *worst;
        (void)hipMalloc(&bad, 8); (void)hipMalloc(&worst, 8); (void)hipMemset(bad, 0, 8); (void)hipMemset(worst, 0, 8);
        check<<<1024, 256>>>(q, xbits, 256, bad, worst);
        unsigned long long hb; double hw;
        (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hw, worst, 8, hipMemcpyDeviceToHost);
        printf("exactness q=%llu (%.1f bits) |x|<2^%d : %llu bad of %d, max |t|/q = %.4f\n", (unsigned long long)q, log2((double)q), xbits, hb,
               1024 * 256 * 256 * 2, hw);
    }
    for (int w : {1, 2, 4}) {
        run<0>("integer asm dual butterfly (shipped, Level B)", w);
        run<1>("f64 error-free, rint", w);
        run<2>("f64 error-free, magic-constant rounding", w);
        run<3>("f64 rint + range reduction once per pass", w);
    }
    return 0;
}
