// What HBM gives a read-dominated streaming kernel on gfx950: S concurrent row streams read with non-temporal 16-byte loads,
// W rows written, the access pattern of the key-switch inner product (per (ciphertext, modulus): L digit rows in, 2 rows out;
// 16 KiB of each row per workgroup).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_stream.hip -o /tmp/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 __attribute__((ext_vector_type(2))) vv;
#define N 32768u
#define CHUNK 2048u

template <int S, int W, int PT, bool STRIDED = false>
__global__ void __launch_bounds__(256) k_stream(const u64 *__restrict__ in, u64 *__restrict__ out, unsigned groups, unsigned rows_in) {
    const unsigned chunks = N / CHUNK, g = blockIdx.x / chunks, chunk = blockIdx.x % chunks;   // g: group of PT "ciphertexts"
    for (unsigned i = chunk * CHUNK + threadIdx.x * 2; i < (chunk + 1) * CHUNK; i += 512) {
        vv acc[PT][W];
#pragma unroll
        for (int c = 0; c < PT; c++)
#pragma unroll
            for (int w = 0; w < W; w++) acc[c][w] = vv{0, 0};
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int c = 0; c < PT; c++) {
                // STRIDED: the engine's digit layout [p][j][k]: item = (k, p) modulus-major, its S rows are (S+1) rows apart
                const unsigned item = g * PT + c, P = 256;
                const size_t row = STRIDED ? (((size_t)(item % P) * S + s) * (S + 1) + (item / P) % (S + 1)) % rows_in
                                           : ((size_t)item * S + s) % rows_in;
                const vv t = __builtin_nontemporal_load(reinterpret_cast<const vv *>(in + row * N + i));
#pragma unroll
                for (int w = 0; w < W; w++) acc[c][w] ^= t + (u64)w;
            }
#pragma unroll
        for (int c = 0; c < PT; c++)
#pragma unroll
            for (int w = 0; w < W; w++) *reinterpret_cast<vv *>(out + ((size_t)(g * PT + c) * W + w) * N + i) = acc[c][w];
    }
}

template <int S, int W, int PT, bool STRIDED = false> void run(const char *name, const u64 *in, u64 *out, unsigned cts, unsigned rows_in) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned groups = cts / PT, blocks = groups * (N / CHUNK);
    k_stream<S, W, PT, STRIDED><<<blocks, 256>>>(in, out, groups, rows_in);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) k_stream<S, W, PT, STRIDED><<<blocks, 256>>>(in, out, groups, rows_in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)cts * (S + W) * N * 8;
    printf("%-44s %7.3f ms  %6.0f GB/s (read %d rows + write %d rows per item, %u items)\n", name, ms, bytes / ms / 1e6, S, W, cts);
}

int main() {
    const unsigned rows_in = 28160, cts = 2816;   // 7.4 GB of input rows, as many items as the C3 inner product has (256 x 11)
    u64 *in, *out;
    if (hipMalloc(&in, (size_t)rows_in * N * 8) != hipSuccess || hipMalloc(&out, (size_t)cts * 6 * N * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 1, (size_t)rows_in * N * 8);
    run<10, 2, 1>("10 streams in, 2 out, 1 item per thread", in, out, cts, rows_in);
    run<10, 2, 4>("10 streams in, 2 out, 4 items per thread", in, out, cts, rows_in);
    run<10, 2, 4, true>("the same, rows strided like [p][j][k]", in, out, cts, rows_in);
    run<10, 2, 1, true>("strided rows, 1 item per thread", in, out, cts, rows_in);
    run<10, 0, 4>("10 streams in, nothing out (4 items)", in, out, cts, rows_in);
    run<1, 1, 4>("copy: 1 in, 1 out (4 items)", in, out, cts * 5, rows_in);
    run<2, 1, 4>("2 in, 1 out (4 items)", in, out, cts * 3, rows_in);
    run<4, 3, 4>("4 in, 3 out (tensor-like, 4 items)", in, out, cts, rows_in);
    return 0;
}
