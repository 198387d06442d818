// VALU cost of the lazy Harvey butterfly on gfx950, isolated from memory: every thread keeps 32
// coefficients in registers and runs 5-stage passes on them in a loop.
// Build: hipcc --offload-arch=gfx950 -O3 -I../hehub_amd/csrc -o ubench_bfly ubench_bfly.hip
#include "hp_device.h"
#include <cstdio>

template <int VARIANT> __global__ void __launch_bounds__(256, 4) kern(u64 *out, const u64 *tw, u64 q, int iters) {
    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; r++) x[r] = out[threadIdx.x + 256 * r];
    const u64 two_q = 2 * q, nq = 0 - q;
    u64 w = tw[threadIdx.x & 7], wh = tw[8 + (threadIdx.x & 7)];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int b = 4; b >= 0; --b) {
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
    for (int w : {1, 2, 4}) { run<0>("C (compiler-selected)", w); run<1>("asm-assisted (nq, mad chains)", w); }
    return 0;
}
