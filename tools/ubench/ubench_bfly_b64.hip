// Can the LEVEL-B butterfly (the reference's exact lazy words, ntt.cpp:160-166) be computed on the FP64 pipe?  For narrow moduli
// (q < 2^44, inputs below 2^50) every lazy value stays below 2^53, and the Harvey word is t = r + q I with r = x w mod q canonical and
// I = [r 2^64 < x rho], rho = w 2^64 mod q -- I = 1 only when r < x rho / 2^64 < 2^27, probability ~2^-13 per butterfly.
//   fast path: level-A product (6) + sign fix + compare of r against a per-limb bound + lo + t, (lo + 2q) - t
//   rare path (wave-uniform branch): the indicator in integer arithmetic
// This file checks the words against hp_butterfly (integer) on the device -- including forced suspects -- and measures cycles per
// wave-butterfly against the shipped integer butterfly (62.5).  Kill criterion for building kernels on it: 1.3 x.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../hehub_amd/csrc -o ubench_bfly_b64 ubench_bfly_b64.hip
#include "hp_device.h"
#include <cstdio>
#include <cmath>

#define FDEV __device__ __forceinline__
FDEV double Dd(u64 v) { return __builtin_bit_cast(double, v); }
FDEV u64 Ud(double d) { return __builtin_bit_cast(u64, d); }

// exact lazy Harvey word of x * w (x < 2^52 integer-valued double, w < q < 2^44), as a double
FDEV double harvey_b64(double x, double w, double u, double q, double bound, u64 rho, u64 qi) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double k = __builtin_rint(x * u);
    const double ta = __builtin_fma(-k, q, h) + l;          // x w - k q in (-q, q)
    double r = ta < 0.0 ? ta + q : ta;                       // canonical residue
    if (__builtin_expect(__any(r < bound), 0)) {             // rare (wave-uniform branch): r < x rho / 2^64 is possible
        const u64 ri = (u64)r, xi = (u64)x;
        const unsigned __int128 lhs = (unsigned __int128)ri << 64, rhs = (unsigned __int128)xi * rho;
        if (lhs < rhs) r += (double)qi;
    }
    return r;
}
FDEV void bfly_b64(double &lo, double &hi, double w, double u, double q, double two_q, double bound, u64 rho, u64 qi) {
    const double t = harvey_b64(hi, w, u, q, bound, rho, qi);
    hi = (lo + two_q) - t;
    lo = lo + t;
}

// the same for a group of G butterflies that share (w, u, rho): one wave-uniform branch per group instead of one per butterfly, so that
// the G independent product chains can be interleaved by the compiler
template <int G> FDEV void bfly_b64_group(double *(&lo)[G], double *(&hi)[G], double w, double u, double q, double two_q, double bound, u64 rho, u64 qi) {
    double r[G];
    bool sus = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double x = *hi[g];
        const double h = x * w;
        const double l = __builtin_fma(x, w, -h);
        const double k = __builtin_rint(x * u);
        const double ta = __builtin_fma(-k, q, h) + l;
        r[g] = ta < 0.0 ? ta + q : ta;
        sus = sus || (r[g] < bound);
    }
    if (__builtin_expect(__any(sus), 0)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u64 ri = (u64)r[g], xi = (u64)*hi[g];
            if (((unsigned __int128)ri << 64) < (unsigned __int128)xi * rho) r[g] += (double)qi;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double t = r[g], a = *lo[g];
        *hi[g] = (a + two_q) - t;
        *lo[g] = a + t;
    }
}

// ---- correctness: words identical to the integer butterfly ------------------------------------------------------------------
__device__ u64 splitmix(u64 &s) {
    u64 z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void check(u64 q, int per_thread, unsigned long long *bad, unsigned long long *suspects, unsigned long long *ones) {
    u64 s = 0x9876543ull * (blockIdx.x * blockDim.x + threadIdx.x + 1) + q;
    const double qd = (double)q, two_q = 2.0 * qd;
    const double bound = (double)(u64)((((unsigned __int128)1 << 51) * q) >> 64) + 2.0;   // x < 2^51: x rho / 2^64 < 2^51 q / 2^64
    for (int i = 0; i < per_thread; ++i) {
        const u64 w = splitmix(s) % q;
        const u64 wh = (u64)(((unsigned __int128)w << 64) / q);
        const u64 rho = (u64)(((unsigned __int128)w << 64) % q);
        u64 x = splitmix(s) & ((1ull << 50) - 1), lo = splitmix(s) & ((1ull << 50) - 1);
        if ((i & 3) == 0 && w) {
            // force a suspect: x = r0 * w^-1 mod q (+ a multiple of q) for a tiny r0, so that x w mod q = r0 < bound (q is prime)
            u64 inv = 1, base = w, e = q - 2;
            while (e) { if (e & 1) inv = (u64)(((unsigned __int128)inv * base) % q); base = (u64)(((unsigned __int128)base * base) % q); e >>= 1; }
            const u64 r0 = splitmix(s) % (u64)bound;
            x = (u64)(((unsigned __int128)r0 * inv) % q) + q * (splitmix(s) % ((1ull << 50) / q));
        }
        // reference words (integer)
        u64 rlo = lo, rhi = x;
        hp_butterfly(rlo, rhi, w, wh, q, 2 * q);
        double dlo = (double)lo, dhi = (double)x;
        const u64 r = (u64)(((unsigned __int128)x * w) % q);
        if ((double)r < bound) atomicAdd(suspects, 1ull);
        if (((unsigned __int128)r << 64) < (unsigned __int128)x * rho) atomicAdd(ones, 1ull);
        bfly_b64(dlo, dhi, (double)w, (double)w / qd, qd, two_q, bound, rho, q);
        if ((u64)dlo != rlo || (u64)dhi != rhi) atomicAdd(bad, 1ull);
    }
}

// ---- timing: the harness of ubench_bfly.hip -----------------------------------------------------------------------------------
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
        const double qd = (double)q, two_q = 2.0 * qd, bound = (double)(u64)((((unsigned __int128)1 << 51) * q) >> 64) + 2.0;
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = (double)(out[threadIdx.x + 256 * r] % q);
        double w = (double)(tw[threadIdx.x & 7] % q), u = w / qd;
        u64 rho = tw[threadIdx.x & 7] % q;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int b = 4; b >= 0; --b) {
                if (VARIANT == 1) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        if (r & (1 << b)) continue;
                        bfly_b64(x[r], x[r | (1 << b)], w, u, qd, two_q, bound, rho, q);
                    }
                } else {
                    constexpr int G = VARIANT == 2 ? 4 : 8;
#pragma unroll
                    for (int i0 = 0; i0 < 16; i0 += G) {
                        double *lo[G], *hi[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const int i = i0 + g, ra = ((i >> b) << (b + 1)) | (i & ((1 << b) - 1));
                            lo[g] = &x[ra]; hi[g] = &x[ra | (1 << b)];
                        }
                        bfly_b64_group<G>(lo, hi, w, u, qd, two_q, bound, rho, q);
                    }
                }
                w += 2.0; u += 1e-13; rho += 3;
            }
            // keep the values inside the range the real transform has (not counted: one instruction per coefficient per pass)
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = x[r] * 0.001 + qd;
        }
#pragma unroll
        for (int r = 0; r < 32; r++) out[threadIdx.x + 256 * r] = (u64)x[r];
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
    printf("%-52s waves/SIMD=%d : %8.3f ms  -> %.1f cycles per wave-butterfly per SIMD (2.1 GHz)\n", name, blocks_per_cu, best,
           best * 1e-3 * 2.1e9 / ((double)iters * 80 * blocks_per_cu));
    (void)hipFree(d); (void)hipFree(tw);
}

int main() {
    for (u64 q : {1099510054913ull, 1099507695617ull, 17592182833153ull /* 44 bits, prime */}) {
        unsigned long long *c; (void)hipMalloc(&c, 24); (void)hipMemset(c, 0, 24);
        check<<<512, 256>>>(q, 512, c, c + 1, c + 2);
        unsigned long long h[3]; (void)hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
        printf("words vs the integer butterfly, q=%llu: %llu bad of %d; suspects (r below the bound) %llu, of which the Harvey word is r + q: %llu\n",
               (unsigned long long)q, h[0], 512 * 256 * 512, h[1], h[2]);
    }
    for (int w : {1, 2, 4}) {
        run<0>("integer asm dual butterfly (shipped, level B)", w);
        run<1>("FP64 level-B-exact butterfly with rare-path fix-up", w);
        run<2>("... one branch per group of 4 butterflies", w);
        run<3>("... one branch per group of 8 butterflies", w);
    }
    return 0;
}
