// hp_api_poly.cpp -- C ABI, part 1: the drop-in host entry points (one call = one reference call), the device-resident
// polynomial batches (transforms, RnsIntVec operators, gathers), the encrypt / decrypt cores and the RNS base conversions.
#include "hp_ctx.h"

#include <cstring>

using namespace hpi;

namespace {

int host_vec(hp_ctx *ctx, int op, uint64_t q, size_t n, const uint64_t *a, const uint64_t *b, uint64_t *out,
             size_t a_words_per_elem) {
    if (q < 2) return fail(ctx, HP_EINVAL, "modulus must be >= 2");
    if (n == 0) return HP_OK;
    if ((op == HP_V_MUL_HYBRID || op == HP_V_MONTGOMERY128) && (q & 1) == 0)
        return fail(ctx, HP_EINVAL, "Montgomery reduction needs an odd modulus");
    hp::ModConsts mc = hp::make_consts(q);
    HpVecConsts c;
    c.q = q; c.mqinv = mc.mqinv; c.r64 = mc.r64; c.r64h = mc.r64h; c.barrett_c = mc.barrett_c;
    hp::u128 c128 = (~(hp::u128)0) / q;
    c.c128_hi = (u64)(c128 >> 64);
    c.c128_lo = (u64)c128;
    const size_t abytes = n * 8 * a_words_per_elem, obytes = n * 8;
    int rc = ws_reserve(ctx, padded(abytes / 8) + 2 * padded(n));
    if (rc) return rc;
    Carver cv(ctx->ws);
    u64 *da = cv.take(abytes / 8), *db = cv.take(n), *dout = cv.take(n);
    HIP_TRY(ctx, hipMemcpyAsync(da, a, abytes, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIP_TRY(ctx, hipMemcpyAsync(db, b, obytes, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "vec");
        if ((rc = chk(ctx, hp_launch_vec(op, c, n, da, db, dout, ctx->stream), "vec kernel"))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(out, dout, obytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}

int host_transform(hp_ctx *ctx, size_t logn, uint64_t q, uint64_t *x, int inverse) {
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    const Plan *plan;
    int rc = get_plan(ctx, logn, &q, 1, true, &plan);
    if (rc) return rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, padded(n)))) return rc;
    u64 *d = (u64 *)ctx->ws;
    HIP_TRY(ctx, hipMemcpyAsync(d, x, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = run_ntt(ctx, batch_job(plan, logn, 1, 1, d, d, 1, 1, inverse, 0)))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(x, d, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}
} // namespace

extern "C" {

// ---- drop-in, host pointers -----------------------------------------------------------
int hp_ntt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t logn, uint64_t q, uint64_t *x) {
    HP_ENTER(ctx);
    return host_transform(ctx, logn, q, x, 0);
}
int hp_intt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t logn, uint64_t q, uint64_t *x) {
    HP_ENTER(ctx);
    return host_transform(ctx, logn, q, x, 1);
}
int hp_cache_ntt_factors_strict(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count) {
    HP_ENTER(ctx);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    for (size_t i = 0; i < count; i++) {
        DevTables t;
        int rc = get_tables(ctx, moduli[i], logn, t);
        if (rc) return rc;
    }
    return HP_OK;
}
int hp_check_chain(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, int montgomery) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli);
    if (logn && !logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, count, logn != 0, &plan);
    if (rc) return rc;
    if (montgomery)
        for (auto &c : plan->consts)
            if ((c.q & 1) == 0) return fail(ctx, HP_EINVAL, "Montgomery reduction needs an odd modulus");
    return HP_OK;
}
int hp_batched_barrett_lazy(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_BARRETT_LAZY, q, n, v, nullptr, v, 1);
}
int hp_batched_barrett(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_BARRETT, q, n, v, nullptr, v, 1);
}
int hp_batched_reduce_strict(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_STRICT, q, n, v, nullptr, v, 1);
}
int hp_batched_mul_mod_hybrid_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *a, const uint64_t *b,
                                   uint64_t *out) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_MUL_HYBRID, q, n, a, b, out, 1);
}
int hp_batched_mul_mod_barrett_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *a, const uint64_t *b,
                                    uint64_t *out) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_MUL_BARRETT, q, n, a, b, out, 1);
}
int hp_batched_montgomery_128_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *in128, uint64_t *out) {
    HP_ENTER(ctx);
    return host_vec(ctx, HP_V_MONTGOMERY128, q, n, in128, nullptr, out, 2);
}

// ---- device-resident batches ---------------------------------------------------------------
int hp_dev_ntt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, d_x);
    HP_ALIGNED(ctx, d_x);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    return run_ntt(ctx, batch_job(plan, logn, L, batch, d_x, d_x, L, L, 0, 0));
}

int hp_dev_intt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x, int strict) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, d_x);
    HP_ALIGNED(ctx, d_x);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    return run_ntt(ctx, batch_job(plan, logn, L, batch, d_x, d_x, L, L, 1, strict));
}

// The transforms as RESIDUES (parity level A as an explicit entry point, whatever the context's level): canonical words through the
// FP64 kernels of hp_ntt_a.hip.  Forward: every output word == ntt.cpp:145-176's word modulo q, in [0, q).  Inverse: the words of
// intt_negacyclic_inplace (ntt.h:88-92 = lazy inverse + reduce_strict).  Input words are lazy words of their limb (below 2 q; checked on the
// device, HP_ERANGE from the next synchronising call); N = 2^11 .. 2^15; every q < 2^50.
static int dev_ntt_residues(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x, int inverse) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, d_x);
    HP_ALIGNED(ctx, d_x);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    bool ok = false;
    if ((rc = ensure_plan_a(ctx, plan, &ok))) return rc;
    if (!ok || ctx->force_generic)
        return fail(ctx, HP_EUNSUPPORTED, "residue transforms need a ring degree of 2^11 .. 2^15 and every modulus below 2^50 (and the tiled kernels enabled)");
    HpNttJob j = batch_job(plan, logn, L, batch, d_x, d_x, L, L, inverse, inverse);
    j.limbs_a = plan->d_limbs_a;
    mark_level_a(ctx);   // (the range guard of the level-A kernels: hp_ctx.cpp range_check)
    return run_ntt(ctx, j);
}
int hp_dev_ntt_residues(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x) {
    return dev_ntt_residues(ctx, logn, L, moduli, batch, d_x, 0);
}
int hp_dev_intt_residues(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x) {
    return dev_ntt_residues(ctx, logn, L, moduli, batch, d_x, 1);
}

static int dev_binary(hp_ctx *ctx, int op, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                      const uint64_t *a, const uint64_t *b, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, a, b, out);
    HP_ALIGNED(ctx, a, b, out);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    if (op == HP_MUL)
        for (auto &c : plan->consts)
            if ((c.q & 1) == 0) return fail(ctx, HP_EINVAL, "Montgomery reduction needs an odd modulus");
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_poly_binary(op, plan->d_limbs, (u32)L, (u32)n, (u32)(batch * L), a, b, out, ctx->stream),
               "poly_binary");
}
int hp_dev_poly_add(hp_ctx *ctx, size_t n, size_t L, const uint64_t *m, size_t batch, const uint64_t *a,
                    const uint64_t *b, uint64_t *out) { return dev_binary(ctx, HP_ADD, n, L, m, batch, a, b, out); }
int hp_dev_poly_sub(hp_ctx *ctx, size_t n, size_t L, const uint64_t *m, size_t batch, const uint64_t *a,
                    const uint64_t *b, uint64_t *out) { return dev_binary(ctx, HP_SUB, n, L, m, batch, a, b, out); }
int hp_dev_poly_mul(hp_ctx *ctx, size_t n, size_t L, const uint64_t *m, size_t batch, const uint64_t *a,
                    const uint64_t *b, uint64_t *out) { return dev_binary(ctx, HP_MUL, n, L, m, batch, a, b, out); }

// a chain of += / -= on polynomials (rns.cpp:58-118) as one pass per 32 terms: the words of the single calls in their order
int hp_dev_poly_fold_rows(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t polys, size_t terms, const uint8_t *negate,
                          const uint64_t *const *d_rows, uint64_t *d_out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, negate, d_rows, d_out);
    HP_ALIGNED(ctx, d_out);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (terms < 1) return fail(ctx, HP_EINVAL, "a chain has at least one term");
    if (polys == 0) return HP_OK;
    for (size_t i = 0; i < polys * terms; i++)
        if (!d_rows[i] || ((uintptr_t)d_rows[i] & 15u)) return fail(ctx, HP_EINVAL, "device rows: NULL or misaligned row");
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    const size_t words = L * n;
    // the running sum is accumulated in place in d_out, polynomial by polynomial, 32 terms per launch: an output row may be its own
    // polynomial's first term (x_0 += ...), nothing else -- a row read by a later launch would have been overwritten by then
    for (size_t p0 = 0; p0 < polys; p0++) {
        const u64 *o = d_out + p0 * words;
        for (size_t i = 0; i < polys * terms; i++) {
            const u64 *r = d_rows[i];
            if (r + words <= o || o + words <= r) continue;
            if (r == o && i == p0 * terms) continue;
            return fail(ctx, HP_EINVAL, "poly_fold_rows: an output row overlaps an input row other than its own first term");
        }
    }
    ProfScope ps(ctx, "elem");
    for (size_t p0 = 0; p0 < polys; p0++) {   // (a chain is long and the polynomials are few: one polynomial's segment per launch keeps the argument block small)
        // segments of at most 32 terms; from the second on, term 0 is the running sum itself (in place: a thread reads what it overwrites)
        size_t done = 0;
        while (done < terms) {
            HpFoldRows fr;
            memset(&fr, 0, sizeof(fr));
            size_t cnt = 0;
            if (done) fr.p[cnt++] = d_out + p0 * words;
            while (done < terms && cnt < HP_FOLD_TERMS_MAX) {
                if (done && negate[done]) fr.neg |= 1u << cnt;
                fr.p[cnt++] = d_rows[p0 * terms + done];
                done++;
            }
            if ((rc = chk(ctx, hp_launch_poly_fold(plan->d_limbs, (u32)L, (u32)n, 1, (u32)cnt, fr, d_out + p0 * words, ctx->stream), "poly_fold")))
                return rc;
        }
    }
    return HP_OK;
}

int hp_dev_poly_scalar_mul(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                           const uint64_t *rns_scalar, const uint64_t *a, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, rns_scalar, a, out);
    HP_ALIGNED(ctx, a, out);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    HpScalars sc;
    memset(&sc, 0, sizeof(sc));
    for (size_t k = 0; k < L; k++) {
        sc.s[k] = rns_scalar[k] % moduli[k];                 // rns.cpp:145,163
        sc.sh[k] = hp::harvey_quotient(sc.s[k], moduli[k]);  // rns.cpp:146,164
    }
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_poly_scalar_mul(plan->d_limbs, sc, (u32)L, (u32)n, (u32)(batch * L), a, out, ctx->stream),
               "poly_scalar_mul");
}

int hp_dev_poly_reduce_strict(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch, uint64_t *x) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, x);
    HP_ALIGNED(ctx, x);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_poly_strict(plan->d_limbs, (u32)L, (u32)n, (u32)(batch * L), x, ctx->stream), "poly_strict");
}

int hp_dev_copy(hp_ctx *ctx, size_t words, const uint64_t *d_src, uint64_t *d_dst) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, d_src, d_dst);
    HP_ALIGNED(ctx, d_src, d_dst);
    if (words == 0 || d_src == d_dst) return HP_OK;
    if (words >> 35) return fail(ctx, HP_EUNSUPPORTED, "hp_dev_copy: at most 2^35 words (256 GiB) per call");
    const uintptr_t a = (uintptr_t)d_src, b = (uintptr_t)d_dst;
    if ((a < b && b < a + words * 8) || (b < a && a < b + words * 8)) return fail(ctx, HP_EINVAL, "hp_dev_copy: ranges overlap");
    ProfScope ps(ctx, "copy");
    return chk(ctx, hp_launch_copy(words, d_src, d_dst, ctx->stream), "copy");
}

int hp_dev_poly_involution(hp_ctx *ctx, size_t logn, size_t L, size_t batch, const uint64_t *in, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, in, out);
    HP_ALIGNED(ctx, in, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    if (in == out) return fail(ctx, HP_EINVAL, "involution cannot run in place");
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_reverse((u32)1 << logn, (u32)(batch * L), in, out, ctx->stream), "involution");
}

int hp_dev_poly_cycle(hp_ctx *ctx, size_t logn, size_t L, size_t batch, size_t step, const uint64_t *in,
                      uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, in, out);
    HP_ALIGNED(ctx, in, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (batch == 0) return HP_OK;
    if (in == out) return fail(ctx, HP_EINVAL, "cycle cannot run in place");
    if (step >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
    const size_t n = (size_t)1 << logn;
    const u32 *perm;
    int rc = get_cycle_perm(ctx, logn, step, &perm);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_gather(perm, (u32)n, (u32)(batch * L), in, out, ctx->stream), "cycle");
}

// ---- either side of the path: encrypt / decrypt cores, RNS base transforms ------------------------
int hp_dev_rlwe_encrypt_core(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, const int64_t *noise,
                             const uint64_t *c1, const uint64_t *pt, const uint64_t *sk, uint64_t *ct) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, noise, c1, pt, sk, ct);
    HP_ALIGNED(ctx, noise, c1, pt, sk, ct);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 1 || L > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "invalid component number");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, padded(batch * L * n)))) return rc;
    Carver cv(ctx->ws);
    u64 *ptn = cv.take(batch * L * n);
    {   // ex = NTT(lift(noise)) into ct[p][0]                          sampling.cpp:77-86
        ProfScope ps(ctx, "elem");
        if ((rc = chk(ctx, hp_launch_lift_noise(plan->d_limbs, (u32)L, (u32)n, (u32)batch, (const long long *)noise, ct, (u32)(2 * L),
                                                ctx->stream), "lift_noise"))) return rc;
    }
    if ((rc = run_ntt(ctx, batch_job(plan, logn, L, batch, ct, ct, 2 * L, 2 * L, 0, 0)))) return rc;
    if ((rc = run_ntt(ctx, batch_job(plan, logn, L, batch, pt, ptn, L, L, 0, 0)))) return rc;   // rlwe.cpp:66-67
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_enc_fin(plan->d_limbs, (u32)L, (u32)n, (u32)batch, c1, sk, ptn, ct, ctx->stream), "enc_fin");
}

int hp_dev_rlwe_decrypt_core(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, const uint64_t *ct,
                             const uint64_t *sk, uint64_t *pt) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, moduli, ct, sk, pt);
    HP_ALIGNED(ctx, ct, sk, pt);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, HP_LOGN_MSG);
    if (L < 1 || L > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "invalid component number");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    const size_t n = (size_t)1 << logn;
    {
        ProfScope ps(ctx, "elem");
        if ((rc = chk(ctx, hp_launch_dec_fma(plan->d_limbs, (u32)L, (u32)n, (u32)batch, ct, sk, pt, ctx->stream), "dec_fma"))) return rc;
    }
    return run_ntt(ctx, batch_job(plan, logn, L, batch, pt, pt, L, L, 1, 1));   // rlwe.cpp:78-80
}

int hp_dev_rns_base_from_single(hp_ctx *ctx, size_t n, uint64_t old_modulus, size_t L, const uint64_t *new_moduli, size_t batch,
                                const uint64_t *in, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, new_moduli, in, out);
    HP_ALIGNED(ctx, in, out);
    if (old_modulus < 2 || L < 1 || L > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "invalid moduli");
    if (batch == 0 || n == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, new_moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_base_from_single(plan->d_limbs, old_modulus, (u32)L, (u32)n, (u32)batch, in, out, ctx->stream),
               "base_from_single");
}

int hp_dev_rns_base_to_single_small(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, uint64_t new_modulus,
                                    size_t batch, const uint64_t *in, uint64_t *out, uint32_t *not_small) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, old_moduli, in, out, not_small);
    HP_ALIGNED(ctx, in, out, not_small);
    if (new_modulus < 2 || L < 1 || L > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "invalid moduli");
    if (batch == 0 || n == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, old_moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_base_to_single(plan->d_limbs, (u32)L, (u32)n, (u32)batch, new_modulus, in, out, not_small, ctx->stream),
               "base_to_single");
}

// rns_transform.h rns_base_transform(poly, {new_modulus}) complete: small-coefficient branch for the polynomials whose
// coefficients are all small, CRT composition for the others -- decided per polynomial on the device, no host round trip
int hp_dev_rns_base_to_single(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, uint64_t new_modulus, size_t batch,
                              const uint64_t *in, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, old_moduli, in, out);
    HP_ALIGNED(ctx, in, out);
    if (new_modulus < 2 || L < 1) return fail(ctx, HP_EINVAL, "invalid moduli");
    if (L > HP_CRT_MAX_LIMBS || new_modulus >> 62) return fail(ctx, HP_EUNSUPPORTED, "CRT branch: at most 16 moduli and a new modulus below 2^62");
    for (size_t a = 0; a < L; a++)
        if (!(old_moduli[a] & 1) || old_moduli[a] < 3) return fail(ctx, HP_EUNSUPPORTED, "CRT branch needs odd moduli");
    if (batch == 0 || n == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, old_moduli, L, false, &plan);
    if (rc) return rc;
    const HpCrtConsts *cc;
    if ((rc = get_crt_consts(ctx, old_moduli, L, new_modulus, &cc))) return rc;
    if ((rc = ws_reserve(ctx, padded((batch + 1) / 2)))) return rc;
    u32 *flags = (u32 *)ctx->ws;
    ProfScope ps(ctx, "elem");
    if ((rc = chk(ctx, hp_launch_base_to_single(plan->d_limbs, (u32)L, (u32)n, (u32)batch, new_modulus, in, out, flags, ctx->stream),
                  "base_to_single")))
        return rc;
    return chk(ctx, hp_launch_base_to_single_crt(plan->d_limbs, cc, (u32)L, (u32)n, (u32)batch, in, out, 1, flags, ctx->stream),
               "base_to_single_crt");
}

// extension (the reference throws "under development" for many -> many, rns_transform.cpp:123): the exact CRT value of
// every coefficient, centred around Q/2 exactly as the many -> one CRT branch centres it, reduced into each new modulus
int hp_dev_rns_base_many_to_many(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, size_t Lnew,
                                 const uint64_t *new_moduli, size_t batch, const uint64_t *in, uint64_t *out) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, old_moduli, new_moduli, in, out);
    HP_ALIGNED(ctx, in, out);
    if (L < 1 || Lnew < 1 || Lnew > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "invalid moduli");
    if (L > HP_CRT_MAX_LIMBS) return fail(ctx, HP_EUNSUPPORTED, "CRT composition: at most 16 old moduli");
    for (size_t a = 0; a < L; a++)
        if (!(old_moduli[a] & 1) || old_moduli[a] < 3) return fail(ctx, HP_EUNSUPPORTED, "CRT composition needs odd moduli");
    for (size_t k = 0; k < Lnew; k++)
        if (new_moduli[k] < 2 || new_moduli[k] >> 62) return fail(ctx, HP_EUNSUPPORTED, "new moduli must be in [2, 2^62)");
    if (batch == 0 || n == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, old_moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    for (size_t k = 0; k < Lnew; k++) {
        const HpCrtConsts *cc;
        if ((rc = get_crt_consts(ctx, old_moduli, L, new_moduli[k], &cc))) return rc;
        if ((rc = chk(ctx, hp_launch_base_to_single_crt(plan->d_limbs, cc, (u32)L, (u32)n, (u32)batch, in, out + k * n, (u32)Lnew,
                                                        nullptr, ctx->stream), "base_many_to_many")))
            return rc;
    }
    return HP_OK;
}

} // extern "C"
