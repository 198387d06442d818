// Dependent-issue latency vs throughput of v_mad_u64_u32 / v_lshl_add_u64 / v_mul_hi_u32 on gfx950:
// CH independent chains per wave, WPS waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 ubench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP 32
template <int OP, int CH> __global__ void kern(u64* out, u32 a0, u32 b0, int iters) {
    u32 a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    u64 acc[4] = {threadIdx.x, a, b, a ^ b};
    u32 t[4] = {a, b, a + b, a ^ b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
                if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) & 3]));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t[c]) : "v"(a));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t[c]) : "v"(a));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + t[0] + t[1] + t[2] + t[3];
}
template <int OP, int CH> void run(const char* name, int wps) {
    const int threads = 64 * 4 * wps;   // one block per CU, wps waves per SIMD
    const int blocks = 256, iters = 4000;
    u64* d; (void)hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<OP, CH><<<blocks, threads>>>(d, 12345, 6789, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<OP, CH><<<blocks, threads>>>(d, 12345, 6789, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_wave_instr = (double)iters * REP * CH;
    double ns_per_instr_per_wave = ms * 1e6 / per_wave_instr;      // time between consecutive instrs of one wave
    double simd_cycles = ms * 1e-3 * 2.1e9 / (per_wave_instr * wps); // SIMD cycles per instruction (assuming 2.1 GHz)
    printf("%-16s chains=%d waves/SIMD=%d : %7.3f ms  %.2f ns/instr/wave  ~%.2f SIMD-cycles/instr\n", name, CH, wps, ms,
           ns_per_instr_per_wave, simd_cycles);
    (void)hipFree(d);
}
int main() {
    run<0, 1>("v_mad_u64_u32", 1); run<0, 2>("v_mad_u64_u32", 1); run<0, 4>("v_mad_u64_u32", 1);
    run<0, 1>("v_mad_u64_u32", 2); run<0, 1>("v_mad_u64_u32", 4); run<0, 2>("v_mad_u64_u32", 4); run<0, 4>("v_mad_u64_u32", 4);
    run<1, 1>("v_lshl_add_u64", 1); run<1, 4>("v_lshl_add_u64", 1); run<1, 1>("v_lshl_add_u64", 4);
    run<2, 1>("v_mul_hi_u32", 1); run<2, 4>("v_mul_hi_u32", 1); run<2, 1>("v_mul_hi_u32", 4);
    run<3, 1>("v_add_u32", 1); run<3, 4>("v_add_u32", 1); run<3, 1>("v_add_u32", 4);
    return 0;
}
