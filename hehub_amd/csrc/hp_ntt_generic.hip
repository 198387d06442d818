// hp_ntt_generic.hip -- simple one-stage-at-a-time negacyclic transforms for any logN in [1,16].
//
// These kernels exist for (a) sizes the tiled kernels do not cover (N < 2048: the reference's
// own tests use N = 8, 16, 128; N = 65536: the largest degree the reference's 16-bit bit reversal allows) and (b) as an on-device cross-check of the tiled kernels
// (hp_ctx_set_force_generic).  One workgroup owns one limb; the limb lives in LDS when it fits
// (N <= 8192) and in the destination buffer otherwise.  Every butterfly is exactly the
// reference's (ntt.cpp:160-166), scheduled by stage with a barrier in between.
#include "hp_kernels.h"
#include "hp_ntt_job.h"

#define GEN_THREADS 256

template <bool IN_LDS>
__global__ void __launch_bounds__(GEN_THREADS) k_ntt_generic(HpNttJob job) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const u32 w = hp_xcd_remap(blockIdx.x, job.W);
    HpItem it;
    if (!hp_decode_item(job, w, it)) return;
    const HpLimb m = job.limbs[it.limb];
    const u32 logn = job.logn, n = 1u << logn;
    u64 *buf = IN_LDS ? smem : it.dst;
    if (IN_LDS || it.src != it.dst) {
        for (u32 i = threadIdx.x; i < n; i += GEN_THREADS) buf[i] = it.src[i];
    }
    __syncthreads();
    const u32 half = n >> 1;
    if (!job.inverse) {
        // ntt.cpp:155-169
        for (u32 s = 1; s <= logn; s++) {
            const u32 lg = logn - s, gap = 1u << lg;
            for (u32 b = threadIdx.x; b < half; b += GEN_THREADS) {
                const u32 blk = b >> lg, j = b & (gap - 1);
                const u32 l = (blk << (lg + 1)) | j;
                const u64x2 tw = m.fwd_ref[(1u << (s - 1)) + blk];
                u64 lo = buf[l], hi = buf[l + gap];
                hp_butterfly(lo, hi, tw.x, tw.y, m.q, m.two_q);
                buf[l] = lo; buf[l + gap] = hi;
            }
            __syncthreads();
        }
        for (u32 i = threadIdx.x; i < n; i += GEN_THREADS) it.dst[i] = hp_shift_fold(buf[i], m.q, m.k, m.fix);
    } else {
        // ntt.cpp:178-223 in the caller's index space: stage s pairs i and i + 2^s, twiddle level s,
        // entry bitrev(i mod 2^s)
        for (u32 s = 0; s < logn; s++) {
            const u32 gap = 1u << s;
            for (u32 b = threadIdx.x; b < half; b += GEN_THREADS) {
                const u32 blk = b >> s, c = b & (gap - 1);
                const u32 l = (blk << (s + 1)) | c;
                const u32 rc = s ? (__brev(c) >> (32 - s)) : 0u;
                const u64x2 tw = m.inv_ref[(gap - 1) + rc];
                u64 lo = buf[l], hi = buf[l + gap];
                hp_butterfly(lo, hi, tw.x, tw.y, m.q, m.two_q);
                buf[l] = lo; buf[l + gap] = hi;
            }
            __syncthreads();
        }
        for (u32 i = threadIdx.x; i < n; i += GEN_THREADS) {
            const u64x2 sc = m.inv_ref[n + i];
            u64 v = hp_harvey_lazy(hp_shift_fold(buf[i], m.q, m.k, m.fix), sc.x, sc.y, m.q);
            if (job.use_post_scalar) v = hp_harvey_lazy(v, job.post_scalar, job.post_scalar_h, m.q);
            if (job.strict) v = hp_strict(v, m.q);
            it.dst[i] = v;
        }
    }
}

hipError_t hp_launch_ntt_generic(const HpNttJob &job, hipStream_t stream) {
    if (job.W == 0) return hipSuccess;
    const size_t bytes = ((size_t)8) << job.logn;
    if (job.logn <= 13) {
        k_ntt_generic<true><<<job.W, GEN_THREADS, bytes, stream>>>(job);
    } else {
        k_ntt_generic<false><<<job.W, GEN_THREADS, 0, stream>>>(job);
    }
    return hipGetLastError();
}
