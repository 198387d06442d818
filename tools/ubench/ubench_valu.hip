// Micro-benchmark: issue rate of the integer instructions the Harvey butterfly is made of.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define REP 64
template <int OP> __global__ void __launch_bounds__(256) kern(u64* out, u32 a0, u32 b0, int iters) {
    u32 a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    u64 acc0 = threadIdx.x, acc1 = a, acc2 = b, acc3 = a ^ b;
    u32 t0 = a, t1 = b, t2 = a + b, t3 = a ^ b;
    double d0 = a, d1 = b, d2 = 1.0 + a, d3 = 2.0 + b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (OP == 0) { // v_mad_u64_u32, 4 independent chains
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 1) { // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t0) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t1) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t2) : "v"(a));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t3) : "v"(a));
            } else if (OP == 2) { // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t0) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t1) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t2) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t3) : "v"(a));
            } else if (OP == 3) { // v_add_u32 (full-rate reference)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(t0) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(t1) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(t2) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(t3) : "v"(a));
            } else if (OP == 4) { // v_lshl_add_u64 (64-bit add)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc1) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc2) : "v"(acc3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc3));
            } else if (OP == 5) { // v_mul_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t0) : "v"(a));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t1) : "v"(a));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t2) : "v"(a));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(t3) : "v"(a));
            } else if (OP == 6) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d3), "v"(d2));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d3), "v"(d2));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d3), "v"(d2));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d3), "v"(d2));
            } else if (OP == 7) { // v_add_co_u32 + v_addc_co_u32 pair
                u32 l0 = (u32)acc0, h0 = (u32)(acc0 >> 32), l1 = (u32)acc1, h1 = (u32)(acc1 >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(l0), "+v"(h0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(l1), "+v"(h1) : "v"(a), "v"(b) : "vcc");
                acc0 = ((u64)h0 << 32) | l0; acc1 = ((u64)h1 << 32) | l1;
            } else if (OP == 8) { // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t0) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t1) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t2) : "v"(a), "v"(b));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t3) : "v"(a), "v"(b));
            } else if (OP == 9) { // v_lshlrev_b64
                asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc0));
                asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc1));
                asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc2));
                asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc3));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + (u64)(d0 + d1) + t0 + t1 + t2 + t3;
}

template <int OP> void run(const char* name, int per_rep) {
    const int blocks = 256 * 8, threads = 256, iters = 6000;
    u64* d; hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<OP><<<blocks, threads>>>(d, 12345, 6789, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<OP><<<blocks, threads>>>(d, 12345, 6789, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * (threads / 64) * iters * REP * per_rep;   // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9 * 1024.0 / winstr;              // assumes 2.4 GHz, 1024 SIMDs
    printf("%-18s %8.3f ms  %.2f Tinstr-lanes/s  ~%.2f cycles/wave-instr/SIMD (at 2.4GHz)\n", name, ms,
           winstr * 64 / (ms * 1e-3) / 1e12, per_simd_cycles);
    hipFree(d);
}

int main() {
    run<3>("v_add_u32", 4);
    run<3>("v_add_u32", 4);
    run<0>("v_mad_u64_u32", 4);
    run<1>("v_mul_lo_u32", 4);
    run<2>("v_mul_hi_u32", 4);
    run<5>("v_mul_u32_u24", 4);
    run<8>("v_mad_u32_u24", 4);
    run<4>("v_lshl_add_u64", 4);
    run<7>("add_co+addc pair", 4);
    run<9>("v_lshlrev_b64", 4);
    run<6>("v_fma_f64", 4);
    run<3>("v_add_u32", 4);
    run<0>("v_mad_u64_u32", 4);
    return 0;
}
