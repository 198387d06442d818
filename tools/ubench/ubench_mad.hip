// Cost of v_mad_u64_u32 forms with distinct operand registers (register-file port pressure) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
#define REP 16
template <int OP> __global__ void __launch_bounds__(256) kern(u64* out, u32 a0, int iters) {
    u32 a[8]; u64 acc[8], c[8];
    for (int i = 0; i < 8; i++) { a[i] = a0 + threadIdx.x * (i + 1); acc[i] = threadIdx.x + i; c[i] = a[i] * 3ull; }
    u64 sd;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(sd) : "v"(a[i]), "v"(a[(i + 3) & 7]));            // acc += a*b
                if (OP == 1) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(acc[i]), "=s"(sd) : "v"(a[i]), "v"(a[(i + 3) & 7]));            // acc = a*b
                if (OP == 2) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(acc[i]), "=s"(sd) : "v"(a[i]), "v"(a[(i + 3) & 7]), "v"(c[(i+1)&7])); // acc = a*b + c (separate dst)
                if (OP == 3) { u32 t; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(a[(i + 3) & 7])); acc[i] ^= t; }
                if (OP == 4) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(acc[i]) : "v"(c[i]), "v"(c[(i + 1) & 7]));
                if (OP == 5) { u32 t; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(a[(i + 3) & 7])); acc[i] ^= t; }
            }
        }
    }
    u64 s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int extra_per_rep) {
    const int blocks = 256 * 4, threads = 256, iters = 3000;   // 4 waves per SIMD
    u64* d; (void)hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<OP><<<blocks, threads>>>(d, 12345, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); kern<OP><<<blocks, threads>>>(d, 12345, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double groups = (double)iters * REP * 8 * 4;   // op-groups per SIMD
    printf("%-44s %7.3f ms  %.2f cycles per group per SIMD (2.1 GHz)%s\n", name, ms, ms * 1e-3 * 2.1e9 / groups, extra_per_rep ? "  [includes 1 v_xor]" : "");
    (void)hipFree(d);
}
int main() {
    run<0>("v_mad_u64_u32 acc += a*b (in place)", 0);
    run<1>("v_mad_u64_u32 acc = a*b + 0", 0);
    run<2>("v_mad_u64_u32 d = a*b + c (4 dwords read)", 0);
    run<3>("v_mul_lo_u32 (+v_xor)", 1);
    run<5>("v_mul_hi_u32 (+v_xor)", 1);
    run<4>("v_lshl_add_u64 d = a + b", 0);
    return 0;
}
