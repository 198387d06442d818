// hp_ntt_job.h -- decoding of a transform work item (shared by the generic and tiled kernels).
#pragma once
#include "hp_kernels.h"

struct HpItem {
    const u64 *src;
    u64 *dst;
    u32 limb;
    u32 poly;   // HP_NTT_BATCH: the polynomial of the item
};

// (digit spread: the diagonal digit k == j is never an item -- it is the untouched NTT-form input limb,
// rgsw.cpp:99-101, read directly by the inner-product kernel)
HP_DEV bool hp_decode_item(const HpNttJob &job, u32 w, HpItem &it) {
    const size_t n = (size_t)1 << job.logn;
    if (job.mode == HP_NTT_BATCH) {
        u32 k = w / job.P, p = w % job.P;
        if (job.pair_moduli) {   // every limb reads the same row (src_kstride 0): groups of G moduli side by side, as in the digit spread
            const u32 G = job.pair_moduli, ng = job.L / G, full = ng * G * job.P;
            if (w < full) { k = G * (w / (G * job.P)) + (w % G); p = (w % (G * job.P)) / G; }
            else { const u32 r = job.L - ng * G, v = w - full; k = ng * G + v % r; p = v / r; }
        }
        it.src = job.src + ((size_t)p * job.src_pstride + (size_t)k * job.src_kstride) * n;
        it.dst = job.dst + ((size_t)p * job.dst_pstride + k) * n;
        it.limb = k;
        it.poly = p;
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
        if (job.pair_moduli) {
            // groups of G consecutive moduli, items of a group ordered (digit j, polynomial p, modulus k): the G workgroups
            // that transform one source row under the group's moduli are neighbours, the row comes from HBM once per group.
            // Full groups [aG, aG+G) below L hold G(L-1)P items (the digit that is a member has G-1 targets); the last
            // group is what is left of the ciphertext moduli plus the special prime L (which is never a digit).
            const u32 L = job.L, P = job.P, G = job.pair_moduli, nfull = L / G, perg = G * (L - 1) * P;
            u32 a, m, ain, r;
            if (w < nfull * perg) { a = (w / perg) * G; r = w % perg; m = G; ain = G; }
            else { a = nfull * G; r = w - nfull * perg; m = L + 1 - a; ain = m - 1; }
            const u32 r1 = P * m * a, r2 = r1 + P * (m - 1) * ain;
            if (r < r1) { j = r / (P * m); const u32 rem = r % (P * m); p = rem / m; k = a + rem % m; }
            else if (r < r2) {
                const u32 rr = r - r1, pm = P * (m - 1);
                j = a + rr / pm;
                const u32 rem = rr % pm, t = a + rem % (m - 1);
                p = rem / (m - 1);
                k = t + (t >= j ? 1u : 0u);
            } else { const u32 rr = r - r2; j = a + ain + rr / (P * m); const u32 rem = rr % (P * m); p = rem / m; k = a + rem % m; }
        } else if (g < job.L * per) {
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
