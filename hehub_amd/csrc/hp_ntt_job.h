// hp_ntt_job.h -- decoding of a transform work item (shared by the generic and tiled kernels).
#pragma once
#include "hp_kernels.h"

struct HpItem {
    const u64 *src;
    u64 *dst;
    u32 limb;
};

// (digit spread: the diagonal digit k == j is never an item -- it is the untouched NTT-form input limb,
// rgsw.cpp:99-101, read directly by the inner-product kernel)
HP_DEV bool hp_decode_item(const HpNttJob &job, u32 w, HpItem &it) {
    const size_t n = (size_t)1 << job.logn;
    if (job.mode == HP_NTT_BATCH) {
        const u32 k = w / job.P, p = w % job.P;
        it.src = job.src + ((size_t)p * job.src_pstride + (size_t)k * job.src_kstride) * n;
        it.dst = job.dst + ((size_t)p * job.dst_pstride + k) * n;
        it.limb = k;
        return true;
    }
    if (job.mode == HP_NTT_SPREAD) {
        // Hole-free numbering, modulus-major, then digit, then polynomial: for k < L the L-1 digits j != k, for k = L
        // (the special prime) all L digits.  g counts from the first item of modulus 0 (k_first shifts a launch that
        // covers only some moduli).  Neighbouring items read source rows L limbs apart and write digit rows
        // L(L+1) limbs apart (measured 3.5 % faster than polynomial-major with adjacent rows).
        const u32 per = (job.L - 1) * job.P;
        const u32 g = job.k_first * per + w;
        u32 k, j, p;
        if (g < job.L * per) {
            k = g / per;
            const u32 r = g % per, jj = r / job.P;
            p = r % job.P;
            j = jj + (jj >= k ? 1u : 0u);
        } else {
            const u32 r = g - job.L * per;
            k = job.L;
            j = r / job.P;
            p = r % job.P;
        }
        const u32 rest = p * job.L + j;
        it.src = job.src + (size_t)rest * n;
        it.dst = job.dst + ((size_t)rest * (job.L + 1) + k) * n;
        it.limb = k;
        return true;
    }
    if (job.mode == HP_NTT_HKS) {
        // modulus-major: a ciphertext modulus m < L is outside nd-1 digits, a special prime outside all nd
        const u32 L = job.L, nd = job.hks_nd, E = job.hks_E;
        const u32 per = (nd - 1) * job.P;
        u32 m, j, p;
        if (w < L * per) {
            m = w / per;
            const u32 r = w % per, jj = r / job.P, own = m / job.hks_alpha;
            p = r % job.P;
            j = jj + (jj >= own ? 1u : 0u);
        } else {
            const u32 r = w - L * per;
            m = L + r / (nd * job.P);
            const u32 rr = r % (nd * job.P);
            j = rr / job.P;
            p = rr % job.P;
        }
        u64 *row = job.dst + (((size_t)p * nd + j) * E + m) * n;
        it.src = row;
        it.dst = row;
        it.limb = m;
        return true;
    }
    return false;
}
