// hp_ntt_job.h -- decoding of a transform work item (shared by the generic and tiled kernels).
#pragma once
#include "hp_kernels.h"

struct HpItem {
    const u64 *src;
    u64 *dst;
    u32 limb;
};

// returns false when the item is a hole (digit spread, k == j: the diagonal digit is the
// untouched NTT-form input limb, rgsw.cpp:99-101, read directly by the inner-product kernel)
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
        const u32 per = job.P * job.L;
        const u32 k = job.k_first + w / per, rest = w % per;   // rest = p*L + j
        const u32 j = rest % job.L;
        if (k == j) return false;
        it.src = job.src + (size_t)rest * n;
        it.dst = job.dst + ((size_t)rest * (job.L + 1) + k) * n;
        it.limb = k;
        return true;
    }
    return false;
}
