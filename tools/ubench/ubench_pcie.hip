// PCIe transfer cost of a ciphertext that lives in 20 separate 256 KiB host blocks (hehub's SmartArray limbs) against one 5 MiB block:
// pageable / registered, one DMA per block / one kernel that writes all blocks through their device-visible addresses.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_pcie ubench_pcie.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Ptrs { u64 *p[32]; };
__global__ void k_scatter(Ptrs dst, const u64 *src, size_t words_per_row) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    const vv *s = reinterpret_cast<const vv *>(src + blockIdx.y * words_per_row);
    vv *d = reinterpret_cast<vv *>(dst.p[blockIdx.y]);
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words_per_row / 2; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_gather(u64 *dst, Ptrs src, size_t words_per_row) {
    typedef u64 __attribute__((ext_vector_type(2))) vv;
    const vv *s = reinterpret_cast<const vv *>(src.p[blockIdx.y]);
    vv *d = reinterpret_cast<vv *>(dst + blockIdx.y * words_per_row);
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words_per_row / 2; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
int main() {
    const int rows = 20; const size_t words = 32768, bytes = words * 8, reps = 50;
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    u64 *dev; (void)hipMalloc(&dev, rows * bytes); (void)hipMemset(dev, 1, rows * bytes);
    std::vector<u64 *> pageable(rows), reg(rows);
    Ptrs dp;
    for (int r = 0; r < rows; r++) {
        pageable[r] = new u64[words]; reg[r] = new u64[words];
        for (size_t i = 0; i < words; i += 512) { pageable[r][i] = i; reg[r][i] = i; }
        if (hipHostRegister(reg[r], bytes, hipHostRegisterPortable) != hipSuccess) { printf("register failed\n"); return 1; }
        void *d; (void)hipHostGetDevicePointer(&d, reg[r], 0); dp.p[r] = (u64 *)d;
    }
    u64 *big; (void)hipHostMalloc(&big, rows * bytes, hipHostMallocPortable);
    auto run = [&](const char *name, auto f) {
        f(); (void)hipStreamSynchronize(st);
        double t0 = now();
        for (size_t i = 0; i < reps; i++) { f(); (void)hipStreamSynchronize(st); }
        double dt = (now() - t0) / reps;
        printf("%-58s %8.1f us  %6.1f GB/s\n", name, dt * 1e6, rows * bytes / dt / 1e9);
    };
    run("d2h 20 x 256 KiB pageable, sync each (round 3)", [&] { for (int r = 0; r < rows; r++) { (void)hipMemcpyAsync(pageable[r], dev + r * words, bytes, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); } });
    run("d2h 20 x 256 KiB registered, async, one sync", [&] { for (int r = 0; r < rows; r++) (void)hipMemcpyAsync(reg[r], dev + r * words, bytes, hipMemcpyDeviceToHost, st); });
    run("d2h 1 x 5 MiB pinned (hipHostMalloc)", [&] { (void)hipMemcpyAsync(big, dev, rows * bytes, hipMemcpyDeviceToHost, st); });
    run("d2h one kernel writing the 20 registered blocks", [&] { k_scatter<<<dim3(16, rows), 256, 0, st>>>(dp, dev, words); });
    run("h2d 20 x 256 KiB pageable, sync each (round 3)", [&] { for (int r = 0; r < rows; r++) { (void)hipMemcpyAsync(dev + r * words, pageable[r], bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); } });
    run("h2d 20 x 256 KiB registered, async, one sync", [&] { for (int r = 0; r < rows; r++) (void)hipMemcpyAsync(dev + r * words, reg[r], bytes, hipMemcpyHostToDevice, st); });
    run("h2d 1 x 5 MiB pinned (hipHostMalloc)", [&] { (void)hipMemcpyAsync(dev, big, rows * bytes, hipMemcpyHostToDevice, st); });
    run("h2d one kernel reading the 20 registered blocks", [&] { k_gather<<<dim3(16, rows), 256, 0, st>>>(dev, dp, words); });
    for (int g : {4, 64}) {
        char nm[96]; snprintf(nm, sizeof nm, "d2h kernel, %d workgroups per block", g);
        run(nm, [&] { k_scatter<<<dim3(g, rows), 256, 0, st>>>(dp, dev, words); });
        snprintf(nm, sizeof nm, "h2d kernel, %d workgroups per block", g);
        run(nm, [&] { k_gather<<<dim3(g, rows), 256, 0, st>>>(dev, dp, words); });
    }
    return 0;
}
