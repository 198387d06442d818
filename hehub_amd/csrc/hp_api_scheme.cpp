// hp_api_scheme.cpp -- C ABI, part 2: the scheme-level pipelines of hehub's hot path composed from the HIP kernels:
// key switch (rgsw.cpp:57-156), drop-last-prime (rescaling.cpp:14-78, mod_switch.cpp:13-78), relinearisation and
// rotations (ckks/arith.cpp:64-93, bgv/arith.cpp:71-79), the fused mult pipelines, and the limb-range stages of the
// limb-sharded multi-GPU mode.
#include "hp_ctx.h"

#include <algorithm>
#include <cstring>

using namespace hpi;

namespace hpi {

// rgsw.cpp:57-156 on a batch.  pt rows: polynomial p at pt + p*pt_pstride limbs.
// workspace: coef [P][L][N], digits [P][L][L+1][N]
size_t ext_prod_ws_words(size_t n, size_t L, size_t P) { return padded(P * L * n) / 8 + padded(P * L * (L + 1) * n) / 8; }

// (i) c[j] = strict(INTT(pt[j])) for the digits j in [j0, j1)                          rgsw.cpp:103-105
// for_spread_a: the rows go to the level-A digit-spread launch of the same call and nowhere else: written as doubles
int ks_coef(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t j0, size_t j1, const u64 *pt,
            size_t pt_pstride, u64 *coef, bool for_spread_a) {
    const size_t n = (size_t)1 << logn;
    HpNttJob j = batch_job(plan, logn, j1 - j0, P, pt + j0 * n, coef + j0 * n, pt_pstride, L, 1, 1);
    j.limbs = plan->d_limbs + j0;
    if (ctx->cur_a) j.limbs_a = plan->d_limbs_a + j0;   // strict either way: the same words at both levels
    j.dst_f64 = (ctx->cur_a && for_spread_a) ? 1u : 0u;
    return run_ntt(ctx, j);
}

} // namespace hpi

namespace {

// Which output moduli k in [k0, k1) may keep their digit rows D[.][k] = NTT_{q_k}(c[j]) in the 48-bit packed format: the rows are
// workspace (written by the digit-spread launch, read by the inner product, never seen by a caller), so the format is free as
// long as every word survives it.  A word is the folded output of the lazy transform (ntt.cpp:171-175): with kb = round(log2 q),
// m = x >> kb and delta = |q - 2^kb| it is x - (m - fix) q, below 2^(kb+1) and not wrapped whenever m * delta < 2^kb.  The
// transform's input is a strict coefficient row (< max_j q_j) and grows by at most 2q per stage, which bounds m.  Only the tiled
// kernels and the multi-ciphertext inner-product kernels know the format.
static u32 spread_pack_mask(const hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t k0, size_t k1) {
    // measured (tools/ab/ab_pack.sh): -13..-16 % on the inner product at every tiled ring degree, -1.5 % on the spread launch at
    // N = 32768.  (While the inner product computed 64-bit row addresses per load the extra loads of the two planes ate the
    // gain below N = 32768; with buffer addressing they cost nothing.)
    if (!tiled_ok(ctx, logn) || (int)logn < ctx->pack48_min_logn || P < 2 || ctx->no_pack48) return 0;
    u64 cmax = 0;
    for (size_t j = 0; j < L; j++) cmax = std::max(cmax, plan->consts[j].q);
    u32 mask = 0;
    for (size_t k = k0; k < k1; k++) {
        const hp::ModConsts &c = plan->consts[k];
        const u32 kb = c.k;
        if (kb < 20 || kb > 46) continue;
        const u64 pow = (u64)1 << kb, delta = c.q >= pow ? c.q - pow : pow - c.q;
        const unsigned __int128 xmax = (unsigned __int128)cmax + (unsigned __int128)(2 * logn + 2) * (2 * c.q);
        const unsigned __int128 m = (xmax >> kb) + 2;
        if (m * delta < pow) mask |= 1u << k;
    }
    return mask;
}

// (ii) + (iii) for the output moduli k in [k0, k1) of q_0..q_{L-1}, p: every digit limb is needed, only the
// owned columns of digits / key / out are touched
// key_L0: number of ciphertext moduli the key was generated for (>= L; its polynomials have key_L0 + 1 limbs)
// strict_coef: coef was produced by ks_coef in this call (rows are strict residues); false for caller-supplied rows, which are
// then not trusted to be below their moduli and the digit rows stay plain u64
int ks_digits_inner(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t k0, size_t k1, const u64 *coef,
                    const u64 *pt, size_t pt_pstride, const u64 *key, size_t key_L0, u64 *out, u64 *digits, bool strict_coef,
                    bool coef_words = false, const u64 *const *keys = nullptr) {
    const size_t n = (size_t)1 << logn;
    int rc;
    // (ii) D[j][k] = NTT_{q_k}(c[j]), k != j                       rgsw.cpp:108-119
    HpNttJob sj;
    memset(&sj, 0, sizeof(sj));
    sj.limbs = plan->d_limbs; sj.src = coef; sj.dst = digits; sj.logn = (u32)logn; sj.L = (u32)L; sj.P = (u32)P;
    // items: (L-1)*P per modulus k < L (the diagonal digit is not transformed), L*P for the special prime k = L
    const size_t n_lo = (k1 < L ? k1 : L) - (k0 < L ? k0 : L);
    sj.k_first = (u32)k0; sj.W = (u32)(n_lo * (L - 1) * P + (k1 > L ? L * P : 0)); sj.mode = HP_NTT_SPREAD;
    sj.pair_moduli = (k0 == 0 && k1 == L + 1 && L >= 2) ? (u32)ctx->spread_group : 0u;
    if (sj.pair_moduli > L) sj.pair_moduli = (u32)L;
    // digit rows of the output moduli whose words are provably below 2^48 cross HBM as 6 bytes per word (HP_PACK48)
    // (keys: every ciphertext with its own key -- that inner product reads plain rows)
    sj.pack_mask = strict_coef && !keys ? spread_pack_mask(ctx, plan, logn, L, P, k0, k1) : 0u;
    // level A: digit rows as residues in [q/2, 3q/2] (the inner product below is the same integer kernel: its u128 sums then differ
    // from rgsw.cpp:126-149's by multiples of q_k, its Montgomery outputs are congruent to the reference's and below 2 q_k).
    // Caller-supplied coefficient rows (limb-range stages) are not known to be below 2^50: level B -- unless the caller vouches
    // (hp_dev_ks_inner_range_strict: the rows were written by hp_dev_ks_coef_range, strict residues).
    if (ctx->cur_a && strict_coef) {
        sj.limbs_a = plan->d_limbs_a;
        sj.src_words = coef_words ? 1u : 0u;   // rows of this call's own ks_coef are doubles; rows a caller hands in are words
        // ... and, where q_k + 2 <= 2^40, as 5 bytes per word (HP_PACK40, hp_device.h): same preconditions as the 48-bit rows
        if (!ctx->no_pack40)
            for (size_t k = k0; k < k1; k++)
                if (((sj.pack_mask >> k) & 1u) && plan->consts[k].q + 2 <= ((u64)1 << 40)) sj.pack40_mask |= 1u << k;
    }
    if ((rc = run_ntt(ctx, sj))) return rc;
    // (iii) u128 inner product + Montgomery                         rgsw.cpp:121-153
    if (keys) {
        ProfScope ps(ctx, "ks_inner");
        for (size_t p0 = 0; p0 < P && !rc; p0 += HP_KEY_TABLE_MAX) {
            const size_t cnt = P - p0 < HP_KEY_TABLE_MAX ? P - p0 : HP_KEY_TABLE_MAX;
            HpKeyTable kt;
            memset(&kt, 0, sizeof(kt));
            for (size_t b = 0; b < cnt; b++) kt.p[b] = keys[p0 + b];
            rc = chk(ctx, hp_launch_ks_inner_many(plan->d_limbs, (u32)L, (u32)k0, (u32)(k1 - k0), (u32)(key_L0 + 1), (u32)n, (u32)cnt,
                                                  digits + p0 * L * (L + 1) * n, pt + p0 * pt_pstride * n, (u32)pt_pstride, kt,
                                                  out + p0 * 2 * (L + 1) * n, ctx->stream), "ks_inner (a key per ciphertext)");
        }
        return rc;
    }
    {
        ProfScope ps(ctx, "ks_inner");
        rc = chk(ctx, hp_launch_ks_inner(plan->d_limbs, (u32)L, (u32)k0, (u32)(k1 - k0), (u32)(key_L0 + 1), (u32)n, (u32)P, digits, pt,
                                         (u32)pt_pstride, key, out, sj.pack_mask, sj.pack40_mask, ctx->stream), "ks_inner");
    }
    return rc;
}

int ext_prod(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, const u64 *pt, size_t pt_pstride,
             const u64 *key, size_t key_L0, u64 *out, Carver &cv, const u64 *const *keys = nullptr) {
    const size_t n = (size_t)1 << logn;
    u64 *coef = cv.take(P * L * n);
    u64 *digits = cv.take(P * L * (L + 1) * n);
    int rc;
    if ((rc = ks_coef(ctx, plan, logn, L, P, 0, L, pt, pt_pstride, coef, true))) return rc;
    return ks_digits_inner(ctx, plan, logn, L, P, 0, L + 1, coef, pt, pt_pstride, key, key_L0, out, digits, true, false, keys);
}

} // namespace

namespace hpi {
// rescaling.cpp:46-75 / mod_switch.cpp:45-77 on P2 polynomials of L limbs (x rows: poly p2 at x + p2*L limbs)
size_t drop_ws_words(size_t n, size_t L, size_t P2) { return padded(P2 * n) / 8 + padded(P2 * (L - 1) * n) / 8; }
} // namespace hpi

namespace {

void make_drop_consts(const Plan *plan, size_t L, bool bgv, u64 t, HpDropConsts &dc) {
    const u64 q_last = plan->consts[L - 1].q;
    memset(&dc, 0, sizeof(dc));
    dc.q_last = q_last;
    dc.half_q_last = q_last / 2;
    dc.bgv = bgv ? 1 : 0;
    for (size_t k = 0; k + 1 < L; k++) {
        const u64 q = plan->consts[k].q;
        dc.r[k] = q_last % q;
        const u64 inv = hp::inverse_mod_prime(q_last, q) % q;
        dc.inv[k] = inv;
        dc.inv_h[k] = hp::harvey_quotient(inv, q);
        if (bgv) {
            dc.t[k] = t % q;
            dc.t_h[k] = hp::harvey_quotient(dc.t[k], q);
            dc.qlt[k] = (q_last % t) % q;
            dc.qlt_h[k] = hp::harvey_quotient(dc.qlt[k], q);
        }
    }
}

// clast[p2] = strict(INTT_{q_last}(x[p2][last]))  (BGV: times t^-1 before the strict reduction):
// a one-limb batch whose rows are the last limbs of the P2 polynomials
int drop_coeffs(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, bool bgv, u64 t, const u64 *x, u64 *clast) {
    const size_t n = (size_t)1 << logn;
    const u64 q_last = plan->consts[L - 1].q;
    HpNttJob lj;
    memset(&lj, 0, sizeof(lj));
    lj.limbs = plan->d_limbs + (L - 1); lj.src = x + (L - 1) * n; lj.dst = clast; lj.logn = (u32)logn; lj.L = 1;
    lj.P = (u32)P2; lj.src_pstride = (u32)L; lj.dst_pstride = 1; lj.src_kstride = 1; lj.W = (u32)P2; lj.mode = HP_NTT_BATCH;
    lj.inverse = 1; lj.strict = 1;
    if (ctx->cur_a) lj.limbs_a = plan->d_limbs_a + (L - 1);
    if (bgv) {
        const u64 s = hp::inverse_mod_prime(t, q_last) % q_last;
        lj.post_scalar = s;
        lj.post_scalar_h = hp::harvey_quotient(s, q_last);
        if (lj.limbs_a) {
            lj.post_scalar = hp::f64_bits((double)s);
            lj.post_scalar_h = hp::f64_bits((double)s / (double)q_last);
        }
        lj.use_post_scalar = 1;
    }
    return run_ntt(ctx, lj);
}

// out[k] = ((x[k] - NTT_k(centre(barrett_k(clast)))) * inv_k) [* (q_last mod t)] [+ addend[k]] for the limbs k in [k0, k1)
// of the L-1 that remain.  rem: workspace of P2*(k1-k0)*n words (unused by the fused tiled path).
int drop_apply(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, size_t k0, size_t k1, const HpDropConsts &dc0,
               const u64 *x, const u64 *clast, bool clast_strict, const u64 *addend, size_t add_poly_stride, size_t add_ct_stride,
               u32 add_mask, u64 *out, u64 *rem) {
    // clast_strict: the caller vouches that every word of clast is below q_last (drop_last: it has just been written by a
    // strict inverse transform).  Rows handed in over the C ABI get the full Barrett reduction, which is right for any u64.
    const size_t n = (size_t)1 << logn, kc = k1 - k0;
    if (kc == 0) return HP_OK;
    // shift everything that is indexed by the limb number to the first limb of the range
    HpDropConsts dc = dc0;
    for (size_t k = 0; k < kc; k++) {
        dc.r[k] = dc0.r[k0 + k]; dc.inv[k] = dc0.inv[k0 + k]; dc.inv_h[k] = dc0.inv_h[k0 + k];
        dc.t[k] = dc0.t[k0 + k]; dc.t_h[k] = dc0.t_h[k0 + k]; dc.qlt[k] = dc0.qlt[k0 + k]; dc.qlt_h[k] = dc0.qlt_h[k0 + k];
    }
    const HpLimb *limbs = plan->d_limbs + k0;
    x += k0 * n;
    out += k0 * n;
    if (addend) addend += k0 * n;
    int rc;
    // tiled sizes: Barrett + centring fused into the remainder NTT's loads, (x - rem)*inv [+ addend] into its stores
    // (not for a launch of a few limbs at level B: three short launches around the SPLIT transform beat one 45 us workgroup per limb)
    if (fused_drop_ok(ctx, logn) && !(split_ok(ctx, logn, kc * P2) && !ctx->cur_a)) {
        HpNttJob fj = batch_job(plan, logn, kc, P2, clast, nullptr, 1, 0, 0, 0);
        fj.limbs = limbs;
        fj.src_kstride = 0;
        fj.pair_moduli = (u32)ctx->drop_group;
        if (fj.pair_moduli > kc) fj.pair_moduli = (u32)kc;
        HpDropArgs da;
        memset(&da, 0, sizeof(da));
        da.dc = dc; da.x = x; da.L = (u32)L; da.addend = addend; da.add_poly_stride = (u32)add_poly_stride;
        da.add_ct_stride = (u32)add_ct_stride; da.add_mask = add_mask; da.out = out; da.out_stride = (u32)(L - 1);
        da.small_rem = clast_strict ? 1 : 0;   // rescaling.cpp:54-58: strict_barrett_{q_k}(c), c < q_last -- one conditional subtraction when q_last <= 2 q_k
        for (size_t k = k0; k < k1; k++)
            if (plan->consts[L - 1].q > 2 * plan->consts[k].q) da.small_rem = 0;
        ProfScope ps(ctx, "ntt_drop");   // its own family: a different kernel (k_ntt_fwd_drop) with 2-3x the bytes of a plain transform
        const bool a_shape = add_mask == 0 || !addend || add_mask == 3u || (add_mask == 1u && !dc.bgv);
        if (ctx->cur_a && clast_strict && a_shape) {
            // level A: the same launch on the FP64 kernel; every constant as the pair of doubles (v, RN(v / q_k))
            fj.limbs_a = plan->d_limbs_a + k0;
            const double ql = (double)dc.q_last;
            da.dc.q_last = hp::f64_bits(ql);
            da.dc.half_q_last = hp::f64_bits((double)dc.half_q_last);
            for (size_t k = 0; k < kc; k++) {
                const double qk = (double)plan->consts[k0 + k].q;
                da.dc.inv[k] = hp::f64_bits((double)dc.inv[k]); da.dc.inv_h[k] = hp::f64_bits((double)dc.inv[k] / qk);
                da.dc.t[k] = hp::f64_bits((double)dc.t[k]); da.dc.t_h[k] = hp::f64_bits((double)dc.t[k] / qk);
                da.dc.qlt[k] = hp::f64_bits((double)dc.qlt[k]); da.dc.qlt_h[k] = hp::f64_bits((double)dc.qlt[k] / qk);
            }
            return chk(ctx, hp_launch_ntt_a_drop(fj, da, ctx->stream), "fused drop NTT (level A)");
        }
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "fused drop NTT");
    }
    if (split_ok(ctx, logn, kc * P2) && !ctx->cur_a && !ctx->no_fused_drop) {
        // a few limbs: the fused drop around the SPLIT transform (two launches of small workgroups; `rem` holds the rows between them)
        HpNttJob sj = batch_job(plan, logn, kc, P2, clast, rem, 1, kc, 0, 0);
        sj.limbs = limbs;
        sj.src_kstride = 0;
        HpDropArgs da;
        memset(&da, 0, sizeof(da));
        da.dc = dc; da.x = x; da.L = (u32)L; da.addend = addend; da.add_poly_stride = (u32)add_poly_stride;
        da.add_ct_stride = (u32)add_ct_stride; da.add_mask = add_mask; da.out = out; da.out_stride = (u32)(L - 1);
        ProfScope ps(ctx, "ntt_drop");
        return chk(ctx, hp_launch_ntt_split_drop(sj, da, ctx->stream), "fused drop NTT (split)");
    }
    {
        ProfScope ps(ctx, "drop_rem");
        if ((rc = chk(ctx, hp_launch_drop_rem(limbs, dc, (u32)kc, (u32)n, (u32)P2, clast, rem, ctx->stream), "drop_rem"))) return rc;
    }
    HpNttJob rj = batch_job(plan, logn, kc, P2, rem, rem, kc, kc, 0, 0);
    rj.limbs = limbs;
    if ((rc = run_ntt(ctx, rj))) return rc;
    {
        ProfScope ps(ctx, "drop_fin");
        rc = chk(ctx, hp_launch_drop_fin(limbs, dc, (u32)L, (u32)kc, (u32)n, (u32)P2, x, rem, addend, (u32)add_poly_stride,
                                         (u32)add_ct_stride, add_mask, out, ctx->stream), "drop_fin");
    }
    return rc;
}

// Level A only: relinearize's mod-down (drop p = q_L, addend on both polynomials: rgsw.cpp / ckks/arith.cpp:64-71) and the rescale / mod
// switch that follows it in a mult (drop q' = q_{L-1}: rescaling.cpp:46-75, mod_switch.cpp:45-77) as ONE transform per output limb:
// ext [P2][L+1] -> out [P2][L-1].  Residues only (hp_ntt_a.hip: DropPre2A has the algebra): the intermediate rows are never formed
// for the limbs k < L - 1, and of the limb L - 1 only the coefficients the second drop needs (one inverse launch, also by linearity).
bool two_drops_ok(const hp_ctx *ctx, size_t logn, size_t L) { return ctx->cur_a && !ctx->no_double_drop && fused_drop_ok(ctx, logn) && L >= 2; }
int drop_two_last_a(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, bool bgv, u64 t1, u64 t2, const u64 *ext,
                    const u64 *addend, size_t add_poly_stride, size_t add_ct_stride, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn;
    HpDropConsts dc1, dc2;
    make_drop_consts(plan, L + 1, bgv, t1, dc1);
    make_drop_consts(plan, L, bgv, t2, dc2);
    u64 *cp = cv.take(P2 * n), *cq = cv.take(P2 * n);
    int rc;
    auto mulmod = [](u64 a, u64 b, u64 q) { return (u64)(((unsigned __int128)a * b) % q); };
    if ((rc = drop_coeffs(ctx, plan, logn, L + 1, P2, bgv, t1, ext, cp))) return rc;
    {
        // cq: the strict coefficients modulo q' of y_{L-1} = A' ext_{L-1} + addend_{L-1} - NTT(K cp) [times t2^-1], which are
        // INTT(A' ext_{L-1} + addend_{L-1}) - K cp: one inverse launch (k_ntt_inv_mix_a), y_{L-1} itself is never formed
        const size_t kl = L - 1;
        const u64 q2 = plan->consts[kl].q;
        u64 A = dc1.inv[kl], K = A;
        if (bgv) {
            A = mulmod(A, dc1.qlt[kl], q2);
            K = mulmod(A, dc1.t[kl], q2);
        }
        HpNttJob lj;
        memset(&lj, 0, sizeof(lj));
        lj.limbs = plan->d_limbs + kl; lj.limbs_a = plan->d_limbs_a + kl; lj.src = ext + kl * n; lj.dst = cq; lj.logn = (u32)logn; lj.L = 1;
        lj.P = (u32)P2; lj.src_pstride = (u32)(L + 1); lj.dst_pstride = 1; lj.src_kstride = 1; lj.W = (u32)P2; lj.mode = HP_NTT_BATCH;
        lj.inverse = 1; lj.strict = 1;
        if (bgv) {
            const u64 s = hp::inverse_mod_prime(t2, q2) % q2;
            lj.post_scalar = hp::f64_bits((double)s);
            lj.post_scalar_h = hp::f64_bits((double)s / (double)q2);
            lj.use_post_scalar = 1;
        }
        HpInvMixArgs mx;
        memset(&mx, 0, sizeof(mx));
        mx.add = addend + kl * n; mx.add_poly_stride = (u32)add_poly_stride; mx.add_ct_stride = (u32)add_ct_stride;
        mx.A = hp::f64_bits((double)A); mx.A_h = hp::f64_bits((double)A / (double)q2);
        mx.K = hp::f64_bits((double)K); mx.K_h = hp::f64_bits((double)K / (double)q2);
        mx.cprev = cp; mx.prev_q = hp::f64_bits((double)dc1.q_last); mx.prev_half = hp::f64_bits((double)dc1.half_q_last);
        ProfScope ps(ctx, "intt");
        if ((rc = chk(ctx, hp_launch_ntt_a_inv_mix(lj, mx, ctx->stream), "inverse NTT of the combined limb (level A)"))) return rc;
    }
    const size_t kc = L - 1;
    if (kc == 0) return HP_OK;
    HpNttJob fj = batch_job(plan, logn, kc, P2, cp, nullptr, 1, 0, 0, 0);
    fj.limbs = plan->d_limbs;
    fj.limbs_a = plan->d_limbs_a;
    fj.src_kstride = 0;
    fj.pair_moduli = (u32)ctx->drop_group;
    if (fj.pair_moduli > kc) fj.pair_moduli = (u32)kc;
    HpDropArgs da;
    memset(&da, 0, sizeof(da));
    da.x = ext; da.L = (u32)(L + 1); da.addend = addend; da.add_poly_stride = (u32)add_poly_stride; da.add_ct_stride = (u32)add_ct_stride;
    da.add_mask = 3; da.out = out; da.out_stride = (u32)(L - 1);
    da.comb = cq;
    da.dc.bgv = bgv ? 1 : 0;
    da.dc.q_last = hp::f64_bits((double)dc1.q_last);
    da.dc.half_q_last = hp::f64_bits((double)dc1.half_q_last);
    da.q2_last = hp::f64_bits((double)dc2.q_last);
    da.half_q2_last = hp::f64_bits((double)dc2.half_q_last);
    for (size_t k = 0; k < kc; k++) {
        const u64 q = plan->consts[k].q;
        const double qd = (double)q;
        // A = p^-1 [(p mod t1)], m = A [t1], m2 = 1 [t2], B = q'^-1 [(q' mod t2)]     (bracketed factors: BGV)
        u64 A = dc1.inv[k], m = A, m2 = 1 % q, B = dc2.inv[k];
        if (bgv) {
            A = mulmod(A, dc1.qlt[k], q);
            m = mulmod(A, dc1.t[k], q);
            m2 = dc2.t[k];
            B = mulmod(B, dc2.qlt[k], q);
        }
        da.dc.inv[k] = hp::f64_bits((double)A); da.dc.inv_h[k] = hp::f64_bits((double)A / qd);
        da.dc.t[k] = hp::f64_bits((double)m); da.dc.t_h[k] = hp::f64_bits((double)m / qd);
        da.comb_mul[k] = hp::f64_bits((double)m2); da.comb_mul_h[k] = hp::f64_bits((double)m2 / qd);
        da.dc.qlt[k] = hp::f64_bits((double)B); da.dc.qlt_h[k] = hp::f64_bits((double)B / qd);
    }
    ProfScope ps(ctx, "ntt_drop");
    return chk(ctx, hp_launch_ntt_a_drop(fj, da, ctx->stream), "fused double drop NTT (level A)");
}

} // namespace

namespace hpi {
int drop_last(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, bool bgv, u64 t, const u64 *x,
              const u64 *addend, size_t add_poly_stride, size_t add_ct_stride, u32 add_mask, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn;
    HpDropConsts dc;
    make_drop_consts(plan, L, bgv, t, dc);
    u64 *clast = cv.take(P2 * n);
    u64 *rem = cv.take(P2 * (L - 1) * n);
    int rc;
    if ((rc = drop_coeffs(ctx, plan, logn, L, P2, bgv, t, x, clast))) return rc;
    return drop_apply(ctx, plan, logn, L, P2, 0, L - 1, dc, x, clast, true, addend, add_poly_stride, add_ct_stride, add_mask, out, rem);
}
} // namespace hpi

namespace {

int check_ext_args(hp_ctx *ctx, size_t logn, size_t L, size_t batch) {
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 1 || L + 1 > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "Invalid component number in RGSW ciphertext.");
    if (batch == 0) return fail(ctx, HP_EINVAL, "empty batch");
    return HP_OK;
}

} // namespace

extern "C" {

int hp_dev_mult_low_level(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                          const uint64_t *ct1, const uint64_t *ct2, uint64_t *quad) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, ct1, ct2, quad);
    HP_ALIGNED(ctx, ct1, ct2, quad);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "tensor");
    return chk(ctx, hp_launch_tensor(plan->d_limbs, (u32)L, 0, (u32)L, (u32)1 << logn, (u32)batch, ct1, ct2, quad, ctx->stream),
               "tensor");
}

static int key_level_ok(hp_ctx *ctx, size_t L, size_t key_L0) {
    if (key_L0 < L || key_L0 + 1 > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "Inconsistent RGSW ciphertext.");
    return HP_OK;
}

static int dev_ext_prod(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                        const uint64_t *pt, const uint64_t *key, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, key, out);
    HP_ALIGNED(ctx, pt, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    LevelScope lvl(ctx, plan);   // this call at parity level A if the context asks for it and the chain allows
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, ext_prod_ws_words(n, L, batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return ext_prod(ctx, plan, logn, L, batch, pt, L, key, key_L0, out, cv);
}
int hp_dev_ext_prod_montgomery(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                               const uint64_t *pt, const uint64_t *key, uint64_t *out) {
    return dev_ext_prod(ctx, logn, L, L, moduli_ext, batch, pt, key, out);
}
int hp_dev_ext_prod_montgomery_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                  const uint64_t *pt, const uint64_t *key, uint64_t *out) {
    return dev_ext_prod(ctx, logn, L, key_L0, moduli_ext, batch, pt, key, out);
}

static int dev_drop_locked(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, bool bgv, uint64_t t, size_t batch,
                           const uint64_t *ct, uint64_t *out) {
    HP_REQUIRE(ctx, moduli, ct, out);
    HP_ALIGNED(ctx, ct, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (bgv && t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    LevelScope lvl(ctx, plan);   // this call at parity level A if the context asks for it and the chain allows
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, drop_ws_words(n, L, 2 * batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return drop_last(ctx, plan, logn, L, 2 * batch, bgv, t, ct, nullptr, 0, 0, 0, out, cv);
}
static int dev_drop(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, bool bgv, uint64_t t, size_t batch,
                    const uint64_t *ct, uint64_t *out) {
    HP_ENTER(ctx);
    return dev_drop_locked(ctx, logn, L, moduli, bgv, t, batch, ct, out);
}
int hp_dev_ckks_rescale(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, const uint64_t *ct,
                        uint64_t *out) { return dev_drop(ctx, logn, L, moduli, false, 0, batch, ct, out); }
// extension (the reference throws "under development" for dropping_primes >= 2, rescaling.cpp:83-85): `drops` successive
// exact one-prime drops; tmp holds the intermediate levels in two alternating halves of batch*2*(L-1)*N words each
// (unused when drops == 1).  One lock for the whole sequence: no other call on the context can get between two drops.
int hp_dev_ckks_rescale_n(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t drops, size_t batch,
                          const uint64_t *ct, uint64_t *tmp, uint64_t *out) {
    HP_ENTER(ctx);
    if (drops < 1 || drops >= L) return fail(ctx, HP_EINVAL, "The number of primes to be dropped is not positive.");
    if (drops > 1 && !tmp) return fail(ctx, HP_EINVAL, "rescale by several primes needs the intermediate buffer");
    const size_t n = (size_t)1 << logn, half = batch * 2 * (L - 1) * n;
    const uint64_t *src = ct;
    for (size_t d = 0; d < drops; d++) {
        uint64_t *dst = (d + 1 == drops) ? out : tmp + (d & 1) * half;
        int rc = dev_drop_locked(ctx, logn, L - d, moduli, false, 0, batch, src, dst);
        if (rc) return rc;
        src = dst;
    }
    return HP_OK;
}
int hp_dev_bgv_mod_switch(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t t, size_t batch,
                          const uint64_t *ct, uint64_t *out) { return dev_drop(ctx, logn, L, moduli, true, t, batch, ct, out); }

// relinearize on a batch: ext_prod(quad[2]) -> drop p -> += quad[0], quad[1]
static int relin_core(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, bool bgv, u64 inner_t,
                      const u64 *quad, const u64 *key, size_t key_L0, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn;
    u64 *ext = cv.take(P * 2 * (L + 1) * n);
    int rc = ext_prod(ctx, plan, logn, L, P, quad + 2 * L * n, 3 * L, key, key_L0, ext, cv);
    if (rc) return rc;
    return drop_last(ctx, plan, logn, L + 1, 2 * P, bgv, inner_t, ext, quad, L, 3 * L, 3, out, cv);
}
static size_t relin_ws_words(size_t n, size_t L, size_t P) {
    return padded(P * 2 * (L + 1) * n) / 8 + ext_prod_ws_words(n, L, P) + drop_ws_words(n, L + 1, 2 * P);
}

static int dev_relin(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, bool bgv, u64 inner_t,
                     size_t batch, const uint64_t *quad, const uint64_t *key, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, quad, key, out);
    HP_ALIGNED(ctx, quad, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (bgv && inner_t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    LevelScope lvl(ctx, plan);   // this call at parity level A if the context asks for it and the chain allows
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, relin_ws_words(n, L, batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return relin_core(ctx, plan, logn, L, batch, bgv, inner_t, quad, key, key_L0, out, cv);
}
int hp_dev_ckks_relinearize(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                            const uint64_t *quad, const uint64_t *key, uint64_t *out) {
    return dev_relin(ctx, logn, L, L, moduli_ext, false, 0, batch, quad, key, out);
}
int hp_dev_ckks_relinearize_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                               const uint64_t *quad, const uint64_t *key, uint64_t *out) {
    return dev_relin(ctx, logn, L, key_L0, moduli_ext, false, 0, batch, quad, key, out);
}
int hp_dev_bgv_relinearize(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t inner_t,
                           size_t batch, const uint64_t *quad, const uint64_t *key, uint64_t *out) {
    return dev_relin(ctx, logn, L, L, moduli_ext, true, inner_t, batch, quad, key, out);
}

// ckks/arith.cpp:75-93: rotate (cycle by `step`) or conjugate (involution) a batch and switch back to the
// original key: moved = gather(ct); ext = ext_prod(moved[1], key); drop p; out[0] += moved[0]
// many != nullptr: ciphertext b is moved by its own step (steps[b]; conj_of[b] != 0: involution) and switched with its own key keys[b]
struct ManyKeys {
    const size_t *steps;
    const unsigned char *conj_of;
    const uint64_t *const *keys;
    const uint64_t *const *polys;   // not NULL: polynomial h of ciphertext b is polys[2 b + h] (u64[L][N]) instead of ct[b][h]
};
static int dev_ckks_automorphism(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                 bool conj, size_t step, const uint64_t *ct, const uint64_t *key, uint64_t *out,
                                 const ManyKeys *many = nullptr) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, out);
    HP_ALIGNED(ctx, out);
    if (!(many && many->polys)) {
        HP_REQUIRE(ctx, ct);
        HP_ALIGNED(ctx, ct);
    }
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (many) {
        HP_REQUIRE(ctx, many->steps, many->keys);
        for (size_t b = 0; many->polys && b < 2 * batch; b++)
            if (!many->polys[b] || ((uintptr_t)many->polys[b] & 15u)) return fail(ctx, HP_EINVAL, "rotate_many: NULL or misaligned polynomial");
        for (size_t b = 0; b < batch; b++) {
            if (!many->keys[b] || ((uintptr_t)many->keys[b] & 15u)) return fail(ctx, HP_EINVAL, "rotate_many: NULL or misaligned key");
            if (!(many->conj_of && many->conj_of[b]) && many->steps[b] >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
        }
    } else {
        HP_REQUIRE(ctx, key);
        HP_ALIGNED(ctx, key);
    }
    if (!conj && step >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    LevelScope lvl(ctx, plan);   // this call at parity level A if the context asks for it and the chain allows
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn;
    const size_t words = padded(batch * 2 * L * n) / 8 + padded(batch * 2 * (L + 1) * n) / 8 + ext_prod_ws_words(n, L, batch) +
                         drop_ws_words(n, L + 1, 2 * batch);
    if ((rc = ws_reserve(ctx, words * 8))) return rc;
    Carver cv(ctx->ws);
    u64 *moved = cv.take(batch * 2 * L * n);
    u64 *ext = cv.take(batch * 2 * (L + 1) * n);
    {
        ProfScope ps(ctx, "elem");
        if (many) {
            // every ciphertext with its own map (and its polynomials possibly anywhere): sources and maps as kernel arguments
            for (size_t b0 = 0; b0 < batch && !rc; b0 += HP_GATHER_TABLE_MAX) {
                const size_t cnt = batch - b0 < HP_GATHER_TABLE_MAX ? batch - b0 : HP_GATHER_TABLE_MAX;
                HpGatherTable gt;
                memset(&gt, 0, sizeof(gt));
                if ((rc = reserve_cycle_perms(ctx, cnt))) break;   // (a miss in a full map cache empties it: not between these)
                for (size_t b = b0; b < b0 + cnt && !rc; b++) {
                    for (size_t h = 0; h < 2; h++) gt.src[b - b0][h] = many->polys ? many->polys[2 * b + h] : ct + (b * 2 + h) * L * n;
                    if (!(many->conj_of && many->conj_of[b])) rc = get_cycle_perm(ctx, logn, many->steps[b], &gt.perm[b - b0]);
                }
                if (!rc) rc = chk(ctx, hp_launch_gather_many(gt, (u32)cnt, (u32)n, (u32)L, moved + b0 * 2 * L * n, ctx->stream), "cycle / involution");
            }
        } else if (conj) {
            rc = chk(ctx, hp_launch_reverse((u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "involution");
        } else {
            const u32 *perm;
            if ((rc = get_cycle_perm(ctx, logn, step, &perm))) return rc;
            rc = chk(ctx, hp_launch_gather(perm, (u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "cycle");
        }
    }
    if (rc) return rc;
    if ((rc = ext_prod(ctx, plan, logn, L, batch, moved + L * n, 2 * L, key, key_L0, ext, cv, many ? many->keys : nullptr))) return rc;
    return drop_last(ctx, plan, logn, L + 1, 2 * batch, false, 0, ext, moved, L, 2 * L, 1, out, cv);
}

// mult_low_level + relinearize + drop q_last, processed in sub-batches so the working set
// (dominated by the L(L+1) digit limbs per ciphertext) stays small
// polys != nullptr: the operands by address, polys[4 b + {0, 1, 2, 3}] = ct1[b][0], ct1[b][1], ct2[b][0], ct2[b][1] (u64[L][N] each)
static int dev_mult(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, bool bgv, u64 t, u64 inner_t, size_t batch,
                    const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key, uint64_t *out, const uint64_t *const *polys = nullptr) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, key, out);
    HP_ALIGNED(ctx, key, out);
    if (polys) {
        for (size_t b = 0; b < 4 * batch; b++)
            if (!polys[b] || ((uintptr_t)polys[b] & 15u)) return fail(ctx, HP_EINVAL, "mult: NULL or misaligned operand polynomial");
    } else {
        HP_REQUIRE(ctx, ct1, ct2);
        HP_ALIGNED(ctx, ct1, ct2);
    }
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (bgv && t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    LevelScope lvl(ctx, plan);   // this call at parity level A if the context asks for it and the chain allows
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn;
    // Sub-batches alternate between two internal streams so that the HBM-bound kernels of one sub-batch (tensor,
    // key-switch inner product) can overlap the multiply-bound transforms of the other.  HP_MULT_CHUNK /
    // This is opt-in (HP_MULT_STREAMS=2, sub-batch HP_MULT_CHUNK, default batch/2): measured +3 % at the C3 shape,
    // but concurrent launches make per-kernel timings (hp_prof_*, rocprofv3) overlap, so the default keeps one
    // stream and one sub-batch and every reported kernel duration is that of a kernel running alone.
    size_t chunk = batch;
    if (ctx->mult_streams >= 2 && batch >= 2) chunk = (batch + 1) / 2;
    if (ctx->mult_chunk > 0) chunk = ctx->mult_chunk < batch ? ctx->mult_chunk : batch;
    const size_t nstreams = (ctx->mult_streams >= 2 && chunk < batch) ? 2 : 1;
    const size_t chunk_words = padded(chunk * 3 * L * n) / 8 + padded(chunk * 2 * L * n) / 8 + relin_ws_words(n, L, chunk) +
                               drop_ws_words(n, L, 2 * chunk);
    if ((rc = ws_reserve(ctx, nstreams * chunk_words * 8))) return rc;
    hipStream_t user = ctx->stream;
    if (nstreams > 1) {
        for (int i = 0; i < 2; i++) {
            if (!ctx->aux[i]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
            if (!ctx->ev_done[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
        }
        if (!ctx->ev_start) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_start, user));
        for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux[i], ctx->ev_start, 0));
    }
    size_t ci = 0;
    for (size_t b0 = 0; b0 < batch; b0 += chunk, ci++) {
        const size_t P = (batch - b0 < chunk) ? batch - b0 : chunk;
        const size_t si = ci % nstreams;
        if (nstreams > 1) ctx->stream = ctx->aux[si];
        Carver cv((char *)ctx->ws + si * chunk_words * 8);
        u64 *quad = cv.take(P * 3 * L * n);
        u64 *lin = cv.take(P * 2 * L * n);
        if (polys) {
            ProfScope ps(ctx, "tensor");
            for (size_t c0 = 0; c0 < P && !rc; c0 += HP_TENSOR_ROWS_MAX) {
                const size_t cnt = P - c0 < HP_TENSOR_ROWS_MAX ? P - c0 : HP_TENSOR_ROWS_MAX;
                HpTensorRows tr;
                memset(&tr, 0, sizeof(tr));
                for (size_t b = 0; b < cnt; b++)
                    for (size_t h = 0; h < 4; h++) tr.p[b][h] = polys[4 * (b0 + c0 + b) + h];
                rc = chk(ctx, hp_launch_tensor_rows(plan->d_limbs, (u32)L, 0, (u32)L, (u32)n, (u32)cnt, tr, quad + c0 * 3 * L * n, ctx->stream),
                         "tensor (operands by address)");
            }
        } else {
            ProfScope ps(ctx, "tensor");
            rc = chk(ctx, hp_launch_tensor(plan->d_limbs, (u32)L, 0, (u32)L, (u32)n, (u32)P, ct1 + b0 * 2 * L * n,
                                           ct2 + b0 * 2 * L * n, quad, ctx->stream), "tensor");
        }
        // the reference's bgv::relinearize runs its inner mod switch with plain_modulus == 1 (bgv.h:32): inner_t = 1
        if (!rc && two_drops_ok(ctx, logn, L)) {
            // level A: relinearize's mod-down and the rescale / mod switch in one transform per output limb (drop_two_last_a)
            u64 *ext = cv.take(P * 2 * (L + 1) * n);
            rc = ext_prod(ctx, plan, logn, L, P, quad + 2 * L * n, 3 * L, key, key_L0, ext, cv);
            if (!rc) rc = drop_two_last_a(ctx, plan, logn, L, 2 * P, bgv, inner_t, t, ext, quad, L, 3 * L, out + b0 * 2 * (L - 1) * n, cv);
            if (rc) break;
            continue;
        }
        if (!rc) rc = relin_core(ctx, plan, logn, L, P, bgv, inner_t, quad, key, key_L0, lin, cv);
        if (!rc) rc = drop_last(ctx, plan, logn, L, 2 * P, bgv, t, lin, nullptr, 0, 0, 0, out + b0 * 2 * (L - 1) * n, cv);
        if (rc) break;
    }
    ctx->stream = user;
    if (nstreams > 1) {
        for (int i = 0; i < 2; i++) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_done[i], ctx->aux[i]));
            HIP_TRY(ctx, hipStreamWaitEvent(user, ctx->ev_done[i], 0));
        }
    }
    return rc;
}
int hp_dev_ckks_rotate(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t step,
                       const uint64_t *ct, const uint64_t *rot_key, uint64_t *out) {
    return dev_ckks_automorphism(ctx, logn, L, L, moduli_ext, batch, false, step, ct, rot_key, out);
}
int hp_dev_ckks_rotate_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                          size_t step, const uint64_t *ct, const uint64_t *rot_key, uint64_t *out) {
    return dev_ckks_automorphism(ctx, logn, L, key_L0, moduli_ext, batch, false, step, ct, rot_key, out);
}
int hp_dev_ckks_rotate_many(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                            const size_t *steps, const unsigned char *conj, const uint64_t *ct, const uint64_t *const *d_keys,
                            uint64_t *out) {
    const ManyKeys many{steps, conj, d_keys, nullptr};
    return dev_ckks_automorphism(ctx, logn, L, key_L0, moduli_ext, batch, false, 0, ct, nullptr, out, &many);
}
int hp_dev_ckks_rotate_many_rows(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                 const size_t *steps, const unsigned char *conj, const uint64_t *const *d_polys,
                                 const uint64_t *const *d_keys, uint64_t *out) {
    if (!ctx) return HP_EINVAL;
    if (!d_polys) { HP_ENTER(ctx); return fail(ctx, HP_EINVAL, "rotate_many: NULL polynomial table"); }
    const ManyKeys many{steps, conj, d_keys, d_polys};
    return dev_ckks_automorphism(ctx, logn, L, key_L0, moduli_ext, batch, false, 0, nullptr, nullptr, out, &many);
}
int hp_dev_ckks_conjugate(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                          const uint64_t *ct, const uint64_t *conj_key, uint64_t *out) {
    return dev_ckks_automorphism(ctx, logn, L, L, moduli_ext, batch, true, 0, ct, conj_key, out);
}
int hp_dev_ckks_conjugate_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                             const uint64_t *ct, const uint64_t *conj_key, uint64_t *out) {
    return dev_ckks_automorphism(ctx, logn, L, key_L0, moduli_ext, batch, true, 0, ct, conj_key, out);
}
int hp_dev_ckks_mult_relin_rescale(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                                   const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key, uint64_t *out) {
    return dev_mult(ctx, logn, L, L, moduli_ext, false, 0, 1, batch, ct1, ct2, key, out);
}
int hp_dev_ckks_mult_relin_rescale_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                                      size_t batch, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key,
                                      uint64_t *out) {
    return dev_mult(ctx, logn, L, key_L0, moduli_ext, false, 0, 1, batch, ct1, ct2, key, out);
}
// the fused pipelines with the operand polynomials by address (a HOST array of 4 device addresses per pair)
int hp_dev_ckks_mult_relin_rescale_rows(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                        const uint64_t *const *d_polys, const uint64_t *key, uint64_t *out) {
    if (!ctx) return HP_EINVAL;
    if (!d_polys) { HP_ENTER(ctx); return fail(ctx, HP_EINVAL, "mult: NULL operand table"); }
    return dev_mult(ctx, logn, L, key_L0, moduli_ext, false, 0, 1, batch, nullptr, nullptr, key, out, d_polys);
}
int hp_dev_bgv_mult_relin_modswitch_rows(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t t, size_t batch,
                                         const uint64_t *const *d_polys, const uint64_t *key, uint64_t *out) {
    if (!ctx) return HP_EINVAL;
    if (!d_polys) { HP_ENTER(ctx); return fail(ctx, HP_EINVAL, "mult: NULL operand table"); }
    return dev_mult(ctx, logn, L, L, moduli_ext, true, t, 1, batch, nullptr, nullptr, key, out, d_polys);
}
int hp_dev_bgv_mult_relin_modswitch(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t t,
                                    size_t batch, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key,
                                    uint64_t *out) {
    return dev_mult(ctx, logn, L, L, moduli_ext, true, t, 1, batch, ct1, ct2, key, out);
}
int hp_dev_bgv_mult_relin_modswitch_t(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t t,
                                      size_t batch, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key,
                                      uint64_t *out) {
    return dev_mult(ctx, logn, L, L, moduli_ext, true, t, t, batch, ct1, ct2, key, out);
}

// ---- limb-range stages (limb-sharded "latency" mode across GPUs) ---------------------------------
static int range_ok(hp_ctx *ctx, size_t lo, size_t hi, size_t limit) {
    if (lo > hi || hi > limit) return fail(ctx, HP_EINVAL, "limb range out of bounds");
    return HP_OK;
}

int hp_dev_mult_low_level_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, size_t k0,
                                size_t k1, const uint64_t *ct1, const uint64_t *ct2, uint64_t *quad) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, ct1, ct2, quad);
    HP_ALIGNED(ctx, ct1, ct2, quad);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    int rc = range_ok(ctx, k0, k1, L);
    if (rc) return rc;
    if (batch == 0 || k0 == k1) return HP_OK;
    const Plan *plan;
    if ((rc = get_plan(ctx, 0, moduli, L, false, &plan))) return rc;
    ProfScope ps(ctx, "tensor");
    return chk(ctx, hp_launch_tensor(plan->d_limbs, (u32)L, (u32)k0, (u32)(k1 - k0), (u32)1 << logn, (u32)batch, ct1, ct2, quad,
                                     ctx->stream), "tensor");
}

int hp_dev_ks_coef_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t j0, size_t j1,
                         const uint64_t *pt, size_t pt_pstride, uint64_t *coef) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, coef);
    HP_ALIGNED(ctx, pt, coef);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = range_ok(ctx, j0, j1, L))) return rc;
    if (j0 == j1) return HP_OK;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    LevelScope lvl(ctx, plan);   // (strict residues at either level: the same words)
    if (lvl.rc) return lvl.rc;
    return ks_coef(ctx, plan, logn, L, batch, j0, j1, pt, pt_pstride, coef);
}

static int ks_inner_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0, size_t k1,
                          const uint64_t *coef, bool coef_strict, const uint64_t *pt, size_t pt_pstride, const uint64_t *key, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, coef, pt, key, out);
    HP_ALIGNED(ctx, coef, pt, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = range_ok(ctx, k0, k1, L + 1))) return rc;
    if (k0 == k1) return HP_OK;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, padded(batch * L * (L + 1) * n)))) return rc;
    Carver cv(ctx->ws);
    u64 *digits = cv.take(batch * L * (L + 1) * n);
    if (!coef_strict) return ks_digits_inner(ctx, plan, logn, L, batch, k0, k1, coef, pt, pt_pstride, key, L, out, digits, false);
    LevelScope lvl(ctx, plan);   // strict rows: packed digit rows, and the FP64 digit spread (from words) when the context is at level A
    if (lvl.rc) return lvl.rc;
    return ks_digits_inner(ctx, plan, logn, L, batch, k0, k1, coef, pt, pt_pstride, key, L, out, digits, true, true);
}
int hp_dev_ks_inner_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0, size_t k1,
                          const uint64_t *coef, const uint64_t *pt, size_t pt_pstride, const uint64_t *key, uint64_t *out) {
    return ks_inner_range(ctx, logn, L, moduli_ext, batch, k0, k1, coef, false, pt, pt_pstride, key, out);
}
int hp_dev_ks_inner_range_strict(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0, size_t k1,
                                 const uint64_t *coef, const uint64_t *pt, size_t pt_pstride, const uint64_t *key, uint64_t *out) {
    return ks_inner_range(ctx, logn, L, moduli_ext, batch, k0, k1, coef, true, pt, pt_pstride, key, out);
}

int hp_dev_drop_coeffs(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                       const uint64_t *x, uint64_t *clast) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, x, clast);
    HP_ALIGNED(ctx, x, clast);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (P2 == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    LevelScope lvl(ctx, plan);   // (strict residues at either level: the same words)
    if (lvl.rc) return lvl.rc;
    return drop_coeffs(ctx, plan, logn, L, P2, plain_modulus != 0, plain_modulus, x, clast);
}

static int drop_apply_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                            size_t k0, size_t k1, const uint64_t *x, const uint64_t *clast, bool clast_strict, const uint64_t *addend,
                            size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, x, clast, out);
    HP_ALIGNED(ctx, x, clast, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    int rc = range_ok(ctx, k0, k1, L - 1);
    if (rc) return rc;
    if (P2 == 0 || k0 == k1) return HP_OK;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli, L, true, &plan))) return rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, padded(P2 * (k1 - k0) * n)))) return rc;
    Carver cv(ctx->ws);
    u64 *rem = cv.take(P2 * (k1 - k0) * n);
    HpDropConsts dc;
    make_drop_consts(plan, L, plain_modulus != 0, plain_modulus, dc);
    // the _strict form follows the context's parity level (canonical residues at level A); the plain form stays at level B
    LevelScope lvl(ctx, clast_strict ? plan : nullptr);
    if (lvl.rc) return lvl.rc;
    return drop_apply(ctx, plan, logn, L, P2, k0, k1, dc, x, clast, clast_strict, addend, add_poly_stride, add_ct_stride, add_mask, out, rem);
}

int hp_dev_drop_apply_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                            size_t k0, size_t k1, const uint64_t *x, const uint64_t *clast, const uint64_t *addend,
                            size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask, uint64_t *out) {
    return drop_apply_range(ctx, logn, L, moduli, plain_modulus, P2, k0, k1, x, clast, false, addend, add_poly_stride, add_ct_stride, add_mask, out);
}
int hp_dev_drop_apply_range_strict(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                                   size_t k0, size_t k1, const uint64_t *x, const uint64_t *clast, const uint64_t *addend,
                                   size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask, uint64_t *out) {
    return drop_apply_range(ctx, logn, L, moduli, plain_modulus, P2, k0, k1, x, clast, true, addend, add_poly_stride, add_ct_stride, add_mask, out);
}

} // extern "C"
