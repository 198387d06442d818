// VALU cost of the lazy Harvey butterfly on gfx950, isolated from memory: every thread keeps 32
// coefficients in registers and runs 5-stage passes on them in a loop.
// Build: hipcc --offload-arch=gfx950 -O3 -I../../hehub_amd/csrc -o ubench_bfly ubench_bfly.hip
#include "hp_device.h"
#include <cstdio>

// two independent butterflies with their instruction streams interleaved by hand
__device__ __forceinline__ void bf2(u64 &lo_a, u64 &hi_a, u64 &lo_b, u64 &hi_b, u64 w, u64 wh, u64 two_q, u32 n0, u32 n1) {
    const u32 p0 = (u32)wh, p1 = (u32)(wh >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u32 ax0 = (u32)hi_a, ax1 = (u32)(hi_a >> 32), bx0 = (u32)hi_b, bx1 = (u32)(hi_b >> 32);
    const u64 aA = (u64)ax1 * p0 + (u64)__umulhi(ax0, p0);
    const u64 bA = (u64)bx1 * p0 + (u64)__umulhi(bx0, p0);
    u64 aB, aG, aE, bB, bG, bE, sd; u32 ac, bc;
    asm("v_mad_u64_u32 %0, vcc, %9, %10, %11\n\t"
        "v_mad_u64_u32 %1, %8, %9, %12, 0\n\t"
        "v_mad_u64_u32 %2, %8, %9, %13, 0\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
        "v_mad_u64_u32 %4, vcc, %15, %10, %16\n\t"
        "v_mad_u64_u32 %2, %8, %14, %12, %2\n\t"
        "v_mad_u64_u32 %5, %8, %15, %12, 0\n\t"
        "v_addc_co_u32_e64 %7, vcc, 0, 0, vcc\n\t"
        "v_mad_u64_u32 %6, %8, %15, %13, 0\n\t"
        "v_mad_u64_u32 %6, %8, %17, %12, %6"
        : "=&v"(aB), "=&v"(aG), "=&v"(aE), "=&v"(ac), "=&v"(bB), "=&v"(bG), "=&v"(bE), "=&v"(bc), "=&s"(sd)
        : "v"(ax0), "v"(p1), "v"(aA), "v"(w0), "v"(w1), "v"(ax1), "v"(bx0), "v"(bA), "v"(bx1)
        : "vcc");
    const u64 aU = ((u64)ac << 32) | (aB >> 32), bU = ((u64)bc << 32) | (bB >> 32);
    const u64 aQ = (u64)ax1 * p1 + aU, bQ = (u64)bx1 * p1 + bU;
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
    const u64 at = ((u64)ath << 32) | (u32)aG, bt = ((u64)bth << 32) | (u32)bG;
    hi_a = lo_a + two_q - at; lo_a = lo_a + at;
    hi_b = lo_b + two_q - bt; lo_b = lo_b + bt;
}

template <int VARIANT> __global__ void __launch_bounds__(256, 4) kern(u64 *out, const u64 *tw, u64 q, int iters) {
    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = out[threadIdx.x + 256 * r];
    const u64 two_q = 2 * q, nq = 0 - q;
    u64 w = tw[threadIdx.x & 7], wh = tw[8 + (threadIdx.x & 7)];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int b = 4; b >= 0; --b) {
            if (VARIANT == 2) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    // i-th index with bit b clear: insert a zero at bit position b
                    const int ra = ((i >> b) << (b + 1)) | (i & ((1 << b) - 1));
                    const int rb = (((i + 1) >> b) << (b + 1)) | ((i + 1) & ((1 << b) - 1));
                    bf2(x[ra], x[ra | (1 << b)], x[rb], x[rb | (1 << b)], w, wh, two_q, (u32)nq, (u32)(nq >> 32));
                }
                w += 2; wh += 3;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if (r & (1 << b)) continue;
                if (VARIANT == 0) hp_butterfly(x[r], x[r | (1 << b)], w, wh, q, two_q);
                else hp_butterfly_nq(x[r], x[r | (1 << b)], w, wh, two_q, (u32)nq, (u32)(nq >> 32));
            }
            w += 2; wh += 3;   // keep the compiler from hoisting anything
        }
    }
#pragma unroll
    for (int r = 0; r < 32; r++) out[threadIdx.x + 256 * r] = x[r];
}

template <int VARIANT> void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 200;
    u64 *d, *tw;
    (void)hipMalloc(&d, 256 * 32 * 8); (void)hipMalloc(&tw, 16 * 8);
    (void)hipMemset(d, 1, 256 * 32 * 8); (void)hipMemset(tw, 3, 16 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<VARIANT><<<blocks, 256>>>(d, tw, 1099510054913ull, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<VARIANT><<<blocks, 256>>>(d, tw, 1099510054913ull, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // each block = 4 waves = 1 wave per SIMD; blocks_per_cu waves per SIMD
    double bf_per_simd = (double)iters * 80 * blocks_per_cu;
    printf("%-28s waves/SIMD=%d : %8.3f ms  -> %.1f cycles per wave-butterfly per SIMD (2.1 GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.1e9 / bf_per_simd);
    (void)hipFree(d); (void)hipFree(tw);
}

int main() {
    for (int w : {1, 2, 4}) { run<0>("C (compiler-selected)", w); run<1>("asm-assisted (nq, mad chains)", w); run<2>("asm, two butterflies interleaved", w); }
    return 0;
}
