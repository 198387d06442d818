// Does a buffer written by one kernel come back from the 256 MiB Infinity Cache when the next kernel reads it?
// write W bytes (kernel A), read them (kernel B), for W = 16 MiB .. 1 GiB; prints the read rate of B.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_mall.hip -o tools/ubench/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef u64 __attribute__((ext_vector_type(2))) vv;
__global__ void __launch_bounds__(256) k_write(vv *p, size_t n, u64 s) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(vv{i + s, i ^ s}, p + i);
}
__global__ void __launch_bounds__(256) k_write_plain(vv *p, size_t n, u64 s) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = vv{i + s, i ^ s};
}
__global__ void __launch_bounds__(256) k_read(const vv *p, size_t n, u64 *out) {
    u64 acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { vv v = __builtin_nontemporal_load(p + i); acc += v.x ^ v.y; }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    const size_t maxb = 2ull << 30;
    vv *buf; u64 *out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 8);
    hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
    for (int plain = 0; plain < 2; ++plain)
    for (size_t mib : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n = mib * (1ull << 20) / 16;
        float wsum = 0, rsum = 0; const int reps = 10;
        for (int r = 0; r < reps + 2; ++r) {
            hipEventRecord(a);
            if (plain) k_write_plain<<<4096, 256>>>(buf, n, r); else k_write<<<4096, 256>>>(buf, n, r);
            hipEventRecord(b);
            k_read<<<4096, 256>>>(buf, n, out);
            hipEventRecord(c);
            hipEventSynchronize(c);
            float w, rd; hipEventElapsedTime(&w, a, b); hipEventElapsedTime(&rd, b, c);
            if (r >= 2) { wsum += w; rsum += rd; }
        }
        printf("%s stores  %5zu MiB: write %.0f GB/s  read-after-write %.0f GB/s\n", plain ? "plain" : "nt   ", mib,
               mib * 1.048576e-3 / (wsum / reps * 1e-3), mib * 1.048576e-3 / (rsum / reps * 1e-3));
    }
    return 0;
}
