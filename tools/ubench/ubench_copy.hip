// Which launch shape gives a plain device-to-device copy the highest rate on gfx950?  (sets COPY_UNROLL / threads of k_copy in hp_elem.hip
// = the engine's measured stream ceiling).   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_copy.hip -o tools/ubench/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef u64 __attribute__((ext_vector_type(2))) vv;
template <int T, int U, bool NT> __global__ void __launch_bounds__(T) k(size_t pairs, const vv *__restrict__ in, vv *__restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (T * U) + threadIdx.x;
    vv v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * T; if (i < pairs) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * T; if (i < pairs) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; } }
}
// grid-stride persistent form
template <int T, int U, bool NT> __global__ void __launch_bounds__(T) kp(size_t pairs, const vv *__restrict__ in, vv *__restrict__ out) {
    for (size_t base = (size_t)blockIdx.x * (T * U) + threadIdx.x; base < pairs; base += (size_t)gridDim.x * (T * U)) {
        vv v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * T; if (i < pairs) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * T; if (i < pairs) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; } }
    }
}
template <class F> static double run(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a); for (int i = 0; i < 20; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
    const size_t bytes = 1ull << 30, pairs = bytes / 16;
    vv *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 1, bytes);
#define R(T, U, NT) { double ms = run([&] { k<T, U, NT><<<(unsigned)((pairs + T * U - 1) / (T * U)), T>>>(pairs, in, out); }); \
    printf("oneshot    T=%4d U=%d %s: %.0f GB/s\n", T, U, NT ? "nt   " : "plain", 2.0 * bytes / (ms * 1e-3) / 1e9); }
#define P(T, U, NT, G) { double ms = run([&] { kp<T, U, NT><<<G, T>>>(pairs, in, out); }); \
    printf("persistent T=%4d U=%d %s grid=%5d: %.0f GB/s\n", T, U, NT ? "nt   " : "plain", G, 2.0 * bytes / (ms * 1e-3) / 1e9); }
    R(256, 2, true) R(256, 4, true) R(256, 8, true) R(512, 4, true) R(512, 8, true) R(1024, 4, true) R(256, 4, false) R(512, 4, false) R(256, 1, true) R(64, 8, true)
    P(256, 4, true, 2048) P(256, 4, true, 4096) P(256, 8, true, 2048) P(512, 4, true, 2048) P(1024, 4, true, 1024) P(256, 4, false, 4096) P(256, 4, true, 8192) P(256, 2, true, 16384)
    double ms = run([&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyDtoD: %.0f GB/s\n", 2.0 * bytes / (ms * 1e-3) / 1e9);
    return 0;
}
