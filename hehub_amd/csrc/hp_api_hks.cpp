// hp_api_hks.cpp -- C ABI, part 3 (extension): hybrid key switch -- digits of several moduli, several special primes;
// kernels in hp_hks.hip.  Exact integer arithmetic throughout (ModUp / ModDown by mixed-radix composition); keys have
// their own format, so results are pinned by an exact integer model and by decryption, not by hehub's words.
#include "hp_ctx.h"

#include <algorithm>
#include <cstring>

using namespace hpi;

static int get_hks_consts(hp_ctx *ctx, const uint64_t *mext, size_t L, size_t k, size_t alpha, const HpHksConsts **out) {
    return contained(ctx, [&] {
        const size_t E = L + k, nd = (L + alpha - 1) / alpha;
        auto key = std::make_pair(std::vector<u64>(mext, mext + E), std::make_pair(k, alpha));
        auto it = ctx->hks.find(key);
        if (it == ctx->hks.end()) {
            if (ctx->hks.size() >= MAX_HKS) {   // bounded cache (an entry is ~150 KB of device memory)
                HIP_TRY(ctx, hipDeviceSynchronize());
                for (auto &kv : ctx->hks) (void)hipFree(kv.second);
                ctx->hks.clear();
            }
            typedef unsigned __int128 u128;
            std::vector<HpHksConsts> hold(1);
            HpHksConsts &c = hold[0];
            memset(&c, 0, sizeof(c));
            c.L = (u32)L; c.k = (u32)k; c.alpha = (u32)alpha; c.nd = (u32)nd; c.E = (u32)E;
            for (size_t d = 0; d < nd; d++) {
                const size_t first = d * alpha, cnt = std::min(alpha, L - first);
                for (size_t a = 0; a < cnt; a++)
                    for (size_t b = 0; b < a; b++) {
                        const u64 qa = mext[first + a], qb = mext[first + b];
                        if (qb % qa == 0) return fail(ctx, HP_EINVAL, "moduli are not pairwise coprime");
                        c.inv[d][b][a] = hp::inverse_mod_prime(qb % qa, qa) % qa;
                        c.inv_h[d][b][a] = hp::harvey_quotient(c.inv[d][b][a], qa);
                    }
                for (size_t m = 0; m < E; m++) {
                    u64 prod = 1 % mext[m];
                    for (size_t a = 0; a < cnt; a++) {
                        c.pref[d][m][a] = prod;
                        c.pref_h[d][m][a] = hp::harvey_quotient(prod, mext[m]);
                        prod = (u64)((u128)prod * (mext[first + a] % mext[m]) % mext[m]);
                    }
                }
            }
            for (size_t i = 0; i < L; i++) {
                u64 pm = 1 % mext[i];
                for (size_t j = 0; j < k; j++) pm = (u64)((u128)pm * (mext[L + j] % mext[i]) % mext[i]);
                if (pm == 0) return fail(ctx, HP_EINVAL, "moduli are not pairwise coprime");
                c.pinv[i] = hp::inverse_mod_prime(pm, mext[i]) % mext[i];
                c.pinv_h[i] = hp::harvey_quotient(c.pinv[i], mext[i]);
            }
            const uint64_t *pm = mext + L;
            for (size_t i = 0; i < L; i++) {   // P mod q_i: needed by the merged ModDown + rescale whatever k is
                u64 prod = 1 % mext[i];
                for (size_t a = 0; a < k; a++) prod = (u64)((u128)prod * (pm[a] % mext[i]) % mext[i]);
                c.p_mod_q[i] = prod;
                c.p_mod_q_h[i] = hp::harvey_quotient(prod, mext[i]);
            }
            if (k <= HP_HKS_MAX_ALPHA) {   // Garner tables of the one-kernel ModDown conversion
                for (size_t a = 0; a < k; a++)
                    for (size_t b = 0; b < a; b++) {
                        if (pm[b] % pm[a] == 0) return fail(ctx, HP_EINVAL, "moduli are not pairwise coprime");
                        c.pg_inv[b][a] = hp::inverse_mod_prime(pm[b] % pm[a], pm[a]) % pm[a];
                        c.pg_inv_h[b][a] = hp::harvey_quotient(c.pg_inv[b][a], pm[a]);
                    }
                for (size_t a = 0; a < k; a++) {   // digits of floor(P/2): residues (p_a - 1)/2
                    u64 u = (pm[a] - 1) / 2;
                    for (size_t b = 0; b < a; b++) u = (u64)((u128)((u + pm[a] - c.p_half[b] % pm[a]) % pm[a]) * c.pg_inv[b][a] % pm[a]);
                    c.p_half[a] = u;
                }
                for (size_t i = 0; i < L; i++) {
                    u64 prod = 1 % mext[i];
                    for (size_t a = 0; a < k; a++) {
                        c.p_pref[i][a] = prod;
                        c.p_pref_h[i][a] = hp::harvey_quotient(prod, mext[i]);
                        prod = (u64)((u128)prod * (pm[a] % mext[i]) % mext[i]);
                    }
                }
            }
            HpHksConsts *d = nullptr;
            int rc = upload(ctx, &c, sizeof(c), (void **)&d);
            if (rc) return rc;
            it = ctx->hks.emplace(key, d).first;
        }
        *out = it->second;
        return (int)HP_OK;
    });
}

static u64 hc_host_pinv(hp_ctx *, const uint64_t *mext, size_t L, size_t k, size_t i, u64 *harvey) {
    typedef unsigned __int128 u128;
    u64 pm = 1 % mext[i];
    for (size_t j = 0; j < k; j++) pm = (u64)((u128)pm * (mext[L + j] % mext[i]) % mext[i]);
    const u64 inv = hp::inverse_mod_prime(pm, mext[i]) % mext[i];
    *harvey = hp::harvey_quotient(inv, mext[i]);
    return inv;
}

static size_t hks_ws_words(size_t n, size_t L, size_t k, size_t nd, size_t P) {
    const size_t E = L + k;
    return padded(P * L * n) / 8 + padded(P * nd * E * n) / 8 + padded(P * 2 * E * n) / 8 + padded(2 * P * k * n) / 8 +
           padded(2 * P * L * n) / 8;
}

// key switch of P polynomials pt (NTT form, L limbs, row stride pt_pstride) with a hybrid key u64[nd][2][L+k][N]:
// out [P][2][L][N] = ModDown( sum_d D_d * key_d ) [+ addend rows (p2>>1)*add_ct_stride + (p2&1)*add_poly_stride + i]
// first part: ks [P][2][E][N] = sum_d D_d * key_d (NTT form) and rem [2P][L][N] = the centred exact conversion of its P-part into
// every q_i (coefficient form)
static int hks_front(hp_ctx *ctx, const Plan *plan, const HpHksConsts *hc, size_t logn, size_t L, size_t k, size_t alpha, size_t P,
                     const u64 *pt, size_t pt_pstride, const u64 *key, const uint64_t *mext, u64 **ks_out, u64 **rem_out, Carver &cv) {
    const size_t n = (size_t)1 << logn, E = L + k, nd = (L + alpha - 1) / alpha;
    u64 *coef = cv.take(P * L * n), *lifted = cv.take(P * nd * E * n), *ks = cv.take(P * 2 * E * n);
    u64 *yp = cv.take(2 * P * k * n), *rem = cv.take(2 * P * L * n);
    *ks_out = ks; *rem_out = rem;
    int rc;
    // coefficients of the input, strictly reduced (as rgsw.cpp:103-105)
    if ((rc = ks_coef(ctx, plan, logn, L, P, 0, L, pt, pt_pstride, coef))) return rc;
    {   // ModUp: every digit's exact integer into every modulus outside the digit
        ProfScope ps(ctx, "hks_modup");
        if ((rc = chk(ctx, hp_launch_hks_modup(plan->d_limbs, hc, (u32)alpha, (u32)nd, (u32)n, (u32)P, coef, lifted, ctx->stream), "hks_modup")))
            return rc;
    }
    {   // transforms of the lifted limbs, in place
        HpNttJob j;
        memset(&j, 0, sizeof(j));
        j.limbs = plan->d_limbs; j.src = lifted; j.dst = lifted; j.logn = (u32)logn; j.L = (u32)L; j.P = (u32)P;
        j.hks_nd = (u32)nd; j.hks_E = (u32)E; j.hks_alpha = (u32)alpha; j.mode = HP_NTT_HKS;
        j.W = (u32)(L * (nd - 1) * P + k * nd * P);
        // parity level A: the lifted rows are canonical residues (ModUp's exact conversion), the inner product takes any
        // representative, and hybrid results have no word-level contract with hehub (other keys): the FP64 transform where allowed
        if (ctx->cur_a) j.limbs_a = plan->d_limbs_a;
        if ((rc = run_ntt(ctx, j))) return rc;
    }
    {
        ProfScope ps(ctx, "ks_inner");
        if ((rc = chk(ctx, hp_launch_hks_inner(plan->d_limbs, (u32)L, (u32)E, (u32)nd, (u32)alpha, (u32)n, (u32)P, lifted, pt, (u32)pt_pstride,
                                               key, ks, ctx->stream), "hks_inner")))
            return rc;
    }
    // ModDown: coefficients of the P-part (strict), centred exact conversion into every q_i, transform, subtract, * P^-1
    {
        HpNttJob j = batch_job(plan, logn, k, 2 * P, ks + L * n, yp, E, k, 1, 1);
        j.limbs = plan->d_limbs + L;
        if (ctx->cur_a) j.limbs_a = plan->d_limbs_a + L;   // (strict either way: the same words)
        if ((rc = run_ntt(ctx, j))) return rc;
    }
    if (k <= HP_HKS_MAX_ALPHA) {
        ProfScope ps(ctx, "hks_moddown");
        if ((rc = chk(ctx, hp_launch_hks_moddown(plan->d_limbs, hc, (u32)k, (u32)n, (u32)(2 * P), yp, rem, ctx->stream), "hks_moddown")))
            return rc;
    } else {   // many special primes: one composition per target modulus (hp_elem.hip)
        const Plan *pplan;
        if ((rc = get_plan(ctx, 0, mext + L, k, false, &pplan))) return rc;
        for (size_t i = 0; i < L; i++) {
            const HpCrtConsts *cc;
            if ((rc = get_crt_consts(ctx, mext + L, k, mext[i], &cc))) return rc;
            ProfScope ps(ctx, "hks_moddown");
            if ((rc = chk(ctx, hp_launch_base_to_single_crt(pplan->d_limbs, cc, (u32)k, (u32)n, (u32)(2 * P), yp, rem + i * n, (u32)L,
                                                            nullptr, ctx->stream), "hks_moddown")))
                return rc;
        }
    }
    return HP_OK;
}

// Parity level A for the drops that end a hybrid key switch (round 6; DESIGN section 7 item 1 of round 5): the FP64 drop kernels of
// hp_ntt_a.hip take them as they are -- their compile-time flavours 1 / 2 / 5 (no addend / addend on both polynomials / on polynomial 0)
// for ModDown, flavour 6 (two drops in one transform) for ModDown merged with the rescale.  What differs from hehub's drops is only
// that the transform's input rows already ARE the per-limb remainders: no centring, which the kernels do against a threshold -- a
// threshold out of reach (2^62) switches it off.  Constants travel as pairs of doubles (v, RN(v / q)) like every level-A constant.
static bool a_drop_shape(const u64 *addend, u32 add_mask) { return !addend || add_mask == 0 || add_mask == 3u || add_mask == 1u; }
static void a_pair(u64 v, u64 q, u64 *bits, u64 *bits_h) {
    *bits = hp::f64_bits((double)v);
    *bits_h = hp::f64_bits((double)v / (double)q);
}
static void a_raw_rows(HpDropArgs &da) {
    da.raw_input = 0;
    da.dc.bgv = 0;
    da.dc.q_last = hp::f64_bits(0.0);
    da.dc.half_q_last = hp::f64_bits(4611686018427387904.0);   // 2^62: no word is ever "above half"
}

static int hks_switch(hp_ctx *ctx, const Plan *plan, const HpHksConsts *hc, size_t logn, size_t L, size_t k, size_t alpha, size_t P,
                      const u64 *pt, size_t pt_pstride, const u64 *key, const u64 *addend, size_t add_poly_stride,
                      size_t add_ct_stride, u32 add_mask, const uint64_t *mext, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn, E = L + k;
    u64 *ks, *rem;
    int rc;
    if ((rc = hks_front(ctx, plan, hc, logn, L, k, alpha, P, pt, pt_pstride, key, mext, &ks, &rem, cv))) return rc;
    // transform of the remainders with the rest of ModDown fused into its stores: out = (x - NTT(rem)) * P^-1 [+ addend]
    if (fused_drop_ok(ctx, logn)) {
        HpNttJob fj = batch_job(plan, logn, L, 2 * P, rem, nullptr, L, 0, 0, 0);
        HpDropArgs da;
        memset(&da, 0, sizeof(da));
        da.raw_input = 1;
        for (size_t i = 0; i < L; i++) { da.dc.inv[i] = hc_host_pinv(ctx, mext, L, k, i, &da.dc.inv_h[i]); }
        da.x = ks; da.L = (u32)E; da.addend = addend; da.add_poly_stride = (u32)add_poly_stride; da.add_ct_stride = (u32)add_ct_stride;
        da.add_mask = addend ? add_mask : 0u; da.out = out; da.out_stride = (u32)L;
        ProfScope ps(ctx, "ntt_drop");
        if (ctx->cur_a && a_drop_shape(addend, add_mask)) {   // out = canonical residues of (ks - NTT(rem)) P^-1 [+ addend]
            fj.limbs_a = plan->d_limbs_a;
            a_raw_rows(da);
            for (size_t i = 0; i < L; i++) a_pair(da.dc.inv[i], mext[i], &da.dc.inv[i], &da.dc.inv_h[i]);
            return chk(ctx, hp_launch_ntt_a_drop(fj, da, ctx->stream), "hks fused ModDown (level A)");
        }
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks fused ModDown");
    }
    if ((rc = run_ntt(ctx, batch_job(plan, logn, L, 2 * P, rem, rem, L, L, 0, 0)))) return rc;
    ProfScope ps(ctx, "hks_down_fin");
    return chk(ctx, hp_launch_hks_down_fin(plan->d_limbs, hc, (u32)L, (u32)n, (u32)(2 * P), ks, rem, addend, (u32)add_poly_stride,
                                           (u32)add_ct_stride, add_mask, out, ctx->stream), "hks_down_fin");
}

static int hks_args_ok(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, size_t batch) {
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 1 || k < 1 || k > HP_CRT_MAX_LIMBS || L + k > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "unsupported number of moduli");
    if (alpha < 1 || alpha > HP_HKS_MAX_ALPHA || (L + alpha - 1) / alpha > HP_HKS_MAX_DIGITS)
        return fail(ctx, HP_EINVAL, "unsupported digit size");
    if (batch == 0) return fail(ctx, HP_EINVAL, "empty batch");
    return HP_OK;
}

extern "C" int hp_dev_hks_switch(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                      const uint64_t *pt, const uint64_t *key, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, key, out);
    HP_ALIGNED(ctx, pt, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, moduli_ext, L, k, alpha, &hc))) return rc;
    LevelScope lvl(ctx, plan);   // level A: the transforms of the lifted digits and the coefficient rows on the FP64 kernels
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn, nd = (L + alpha - 1) / alpha;
    if ((rc = ws_reserve(ctx, hks_ws_words(n, L, k, nd, batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return hks_switch(ctx, plan, hc, logn, L, k, alpha, batch, pt, L, key, nullptr, 0, 0, 0, moduli_ext, out, cv);
}

// ckks rotate / conjugate with a hybrid key: moved = gather(ct); out = hks_switch(moved[1]); out[0] += moved[0]
static int dev_hks_automorphism(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *mext, size_t batch,
                                bool conj, size_t step, const uint64_t *ct, const uint64_t *key, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, mext, ct, key, out);
    HP_ALIGNED(ctx, ct, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    if (!conj && step >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, mext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, mext, L, k, alpha, &hc))) return rc;
    LevelScope lvl(ctx, plan);
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn, nd = (L + alpha - 1) / alpha;
    if ((rc = ws_reserve(ctx, (padded(batch * 2 * L * n) / 8 + hks_ws_words(n, L, k, nd, batch)) * 8))) return rc;
    Carver cv(ctx->ws);
    u64 *moved = cv.take(batch * 2 * L * n);
    {
        ProfScope ps(ctx, "elem");
        if (conj) {
            rc = chk(ctx, hp_launch_reverse((u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "involution");
        } else {
            const u32 *perm;
            if ((rc = get_cycle_perm(ctx, logn, step, &perm))) return rc;
            rc = chk(ctx, hp_launch_gather(perm, (u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "cycle");
        }
        if (rc) return rc;
    }
    return hks_switch(ctx, plan, hc, logn, L, k, alpha, batch, moved + L * n, 2 * L, key, moved, L, 2 * L, 1, mext, out, cv);
}
extern "C" int hp_dev_ckks_rotate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                           size_t step, const uint64_t *ct, const uint64_t *rot_key, uint64_t *out) {
    return dev_hks_automorphism(ctx, logn, L, k, alpha, moduli_ext, batch, false, step, ct, rot_key, out);
}
extern "C" int hp_dev_ckks_conjugate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                              const uint64_t *ct, const uint64_t *conj_key, uint64_t *out) {
    return dev_hks_automorphism(ctx, logn, L, k, alpha, moduli_ext, batch, true, 0, ct, conj_key, out);
}

// ckks::mult_low_level + relinearisation with a hybrid key + rescale by the last ciphertext modulus
extern "C" int hp_dev_ckks_mult_relin_rescale_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext,
                                       size_t batch, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key,
                                       uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli_ext, ct1, ct2, key, out);
    HP_ALIGNED(ctx, ct1, ct2, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, moduli_ext, L, k, alpha, &hc))) return rc;
    LevelScope lvl(ctx, plan);
    if (lvl.rc) return lvl.rc;
    const size_t n = (size_t)1 << logn, nd = (L + alpha - 1) / alpha;
    const size_t words = padded(batch * 3 * L * n) / 8 + padded(batch * 2 * L * n) / 8 + hks_ws_words(n, L, k, nd, batch) +
                         drop_ws_words(n, L, 2 * batch) + 2 * (padded(2 * batch * n) / 8);
    if ((rc = ws_reserve(ctx, words * 8))) return rc;
    Carver cv(ctx->ws);
    u64 *quad = cv.take(batch * 3 * L * n), *lin = cv.take(batch * 2 * L * n);
    {
        ProfScope ps(ctx, "tensor");
        if ((rc = chk(ctx, hp_launch_tensor(plan->d_limbs, (u32)L, 0, (u32)L, (u32)n, (u32)batch, ct1, ct2, quad, ctx->stream), "tensor")))
            return rc;
    }
    if (fused_drop_ok(ctx, logn) && !ctx->hks_two_step) {
        // ModDown and the rescale in ONE transform per remaining limb.  With c = the coefficients of the relinearised limb L-1,
        //   ((ks_i - NTT(rem_i)) P^-1 + quad_i - NTT(centre_i(c))) q_last^-1 = ((ks_i - NTT(rem_i + P centre_i(c))) P^-1 + quad_i) q_last^-1
        // so: ModDown of limb L-1 alone -> its coefficients -> rem_i += P centre_i(c) -> one fused transform over limbs 0..L-2.
        // The same residues as the two-step composition below (another lazy representative of them).
        const size_t P2 = 2 * batch, E = L + k;
        u64 *ks, *rem;
        if ((rc = hks_front(ctx, plan, hc, logn, L, k, alpha, batch, quad + 2 * L * n, 3 * L, key, moduli_ext, &ks, &rem, cv))) return rc;
        u64 *r_last = cv.take(P2 * n), *c_last = cv.take(P2 * n);
        HpDropArgs da;
        {
            HpNttJob fj = batch_job(plan, logn, 1, P2, rem + (L - 1) * n, nullptr, L, 0, 0, 0);
            fj.limbs = plan->d_limbs + (L - 1);
            memset(&da, 0, sizeof(da));
            da.raw_input = 1;
            da.dc.inv[0] = hc_host_pinv(ctx, moduli_ext, L, k, L - 1, &da.dc.inv_h[0]);
            da.x = ks + (L - 1) * n; da.L = (u32)E; da.addend = quad + (L - 1) * n; da.add_poly_stride = (u32)L;
            da.add_ct_stride = (u32)(3 * L); da.add_mask = 3u; da.out = r_last; da.out_stride = 1;
            ProfScope ps(ctx, "ntt_drop");
            if (ctx->cur_a) {
                fj.limbs_a = plan->d_limbs_a + (L - 1);
                a_raw_rows(da);
                a_pair(da.dc.inv[0], moduli_ext[L - 1], &da.dc.inv[0], &da.dc.inv_h[0]);
                rc = chk(ctx, hp_launch_ntt_a_drop(fj, da, ctx->stream), "hks ModDown of the last limb (level A)");
            } else {
                rc = chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks ModDown of the last limb");
            }
            if (rc) return rc;
        }
        {
            HpNttJob lj = batch_job(plan, logn, 1, P2, r_last, c_last, 1, 1, 1, 1);
            lj.limbs = plan->d_limbs + (L - 1);
            if (ctx->cur_a) lj.limbs_a = plan->d_limbs_a + (L - 1);
            if ((rc = run_ntt(ctx, lj))) return rc;
        }
        const bool in_loads = !ctx->hks_combine_kernel;   // HP_HKS_COMBINE_KERNEL: the combination as its own kernel
        if (!in_loads) {
            ProfScope ps(ctx, "hks_combine");
            if ((rc = chk(ctx, hp_launch_hks_combine(plan->d_limbs, hc, (u32)L, (u32)n, (u32)P2, c_last, rem, ctx->stream), "hks_combine")))
                return rc;
        }
        HpNttJob fj = batch_job(plan, logn, L - 1, P2, rem, nullptr, L, 0, 0, 0);
        memset(&da, 0, sizeof(da));
        da.raw_input = 1;
        da.fin_on = 1;
        const u64 q_last = moduli_ext[L - 1];
        if (in_loads) { da.comb = c_last; da.comb_half = q_last / 2; }
        for (size_t i = 0; i + 1 < L; i++) {
            u64 pm = 1 % moduli_ext[i];
            for (size_t j = 0; j < k; j++) pm = (u64)((unsigned __int128)pm * (moduli_ext[L + j] % moduli_ext[i]) % moduli_ext[i]);
            da.comb_mul[i] = pm; da.comb_mul_h[i] = hp::harvey_quotient(pm, moduli_ext[i]); da.comb_r[i] = q_last % moduli_ext[i];
            da.dc.inv[i] = hc_host_pinv(ctx, moduli_ext, L, k, i, &da.dc.inv_h[i]);
            da.fin[i] = hp::inverse_mod_prime(q_last % moduli_ext[i], moduli_ext[i]) % moduli_ext[i];
            da.fin_h[i] = hp::harvey_quotient(da.fin[i], moduli_ext[i]);
        }
        da.x = ks; da.L = (u32)E; da.addend = quad; da.add_poly_stride = (u32)L; da.add_ct_stride = (u32)(3 * L); da.add_mask = 3u;
        da.out = out; da.out_stride = (u32)(L - 1);
        ProfScope ps(ctx, "ntt_drop");
        if (ctx->cur_a && in_loads) {
            // the two-drops flavour of hp_ntt_a.hip (DropPre2A has the algebra): z = ((A x + a) - NTT(m c1 + c2)) B with x = ks, a = quad,
            // c1 = the remainders (as they are), c2 = the centred coefficients of the relinearised last limb, A = m = P^-1, B = q_last^-1 --
            // by linearity the level-B line above: ((ks - NTT(rem + P centre(c))) P^-1 + quad) q_last^-1
            fj.limbs_a = plan->d_limbs_a;
            a_raw_rows(da);
            da.fin_on = 0;
            da.q2_last = hp::f64_bits((double)q_last);
            da.half_q2_last = hp::f64_bits((double)(q_last / 2));
            for (size_t i = 0; i + 1 < L; i++) {
                const u64 q = moduli_ext[i], A = da.dc.inv[i], B = da.fin[i];
                a_pair(A, q, &da.dc.inv[i], &da.dc.inv_h[i]);
                a_pair(A, q, &da.dc.t[i], &da.dc.t_h[i]);
                a_pair(1 % q, q, &da.comb_mul[i], &da.comb_mul_h[i]);
                a_pair(B, q, &da.dc.qlt[i], &da.dc.qlt_h[i]);
            }
            return chk(ctx, hp_launch_ntt_a_drop(fj, da, ctx->stream), "hks fused ModDown + rescale (level A)");
        }
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks fused ModDown + rescale");
    }
    if ((rc = hks_switch(ctx, plan, hc, logn, L, k, alpha, batch, quad + 2 * L * n, 3 * L, key, quad, L, 3 * L, 3, moduli_ext, lin, cv)))
        return rc;
    return drop_last(ctx, plan, logn, L, 2 * batch, false, 0, lin, nullptr, 0, 0, 0, out, cv);
}

