// hp_engine.cpp -- the C ABI of include/hehub_amd.h: engine context, table / plan caches,
// argument checks that mirror hehub's exceptions, and the composition of the HIP kernels into
// hehub's key-switch / rescale / mult pipelines.
//
// There is NO CPU fallback in this file: every entry point launches HIP kernels or fails.
#include "../../include/hehub_amd.h"

#include "hp_kernels.h"
#include "hp_tables.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <new>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {

struct DevTables {
    u64x2 *fwd_ref = nullptr, *inv_ref = nullptr, *fwd_k = nullptr, *inv_k = nullptr;
};

struct Plan {
    HpLimb *d_limbs = nullptr;
    std::vector<hp::ModConsts> consts;
};

struct ProfEvent {
    hipEvent_t a, b;
};

} // namespace

struct hp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    bool force_generic = false;
    std::map<std::pair<u64, size_t>, DevTables> tables;          // (q, logn)
    std::map<std::pair<size_t, std::vector<u64>>, Plan> plans;   // (logn, moduli); logn == 0: no transforms needed
    std::map<std::pair<size_t, size_t>, u32 *> perms;            // (logn, step) -> gather map
    std::map<std::pair<std::vector<u64>, u64>, HpCrtConsts *> crt;   // (old moduli, new modulus) -> CRT-branch constants
    std::map<std::pair<std::vector<u64>, std::pair<size_t, size_t>>, HpHksConsts *> hks;   // (extended moduli, (k, alpha))
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // profiling
    std::string prof_family;
    bool prof_on = false;
    std::vector<ProfEvent> prof_events;
    std::vector<hipEvent_t> event_pool;
    // two auxiliary streams for software-pipelined sub-batches (dev_mult)
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_start = nullptr, ev_done[2] = {nullptr, nullptr};
    // tuning / A-B knobs, read from the environment once when the context is created
    int drop_group = 2;           // HP_DROP_GROUP=G: the same numbering for the fused drop launch (every limb reads one coefficient row)
    int spread_group = 2;         // HP_SPREAD_GROUP=G: digit-spread launch numbered by groups of G moduli (0: modulus-major)
    bool hks_two_step = false;    // HP_HKS_TWO_STEP: hybrid mult = switch, then a separate rescale (instead of the merged transform)
    bool no_fused_drop = false;   // HP_NO_FUSED_DROP: separate drop_rem / NTT / drop_fin launches
    int mult_streams = 1;         // HP_MULT_STREAMS=2: software-pipeline two sub-batches in dev_mult
    size_t mult_chunk = 0;        // HP_MULT_CHUNK: sub-batch size of dev_mult (0 = whole batch, or half with 2 streams)
};

namespace {

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                    \
            return HP_EHIP;                                                                     \
        }                                                                                       \
    } while (0)

int fail(hp_ctx *ctx, int code, const std::string &msg) {
    ctx->err = msg;
    return code;
}

// Host-side containers (table / plan / gather-map caches, profiling lists) may throw; nothing may cross the C ABI.
template <class F> int contained(hp_ctx *ctx, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(ctx, HP_ENOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(ctx, HP_ELOGIC, e.what());
    }
}

// a NULL operand would fault on the device and take the process down: reject it at the boundary
inline bool any_null(std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if (!p) return true;
    return false;
}
#define HP_REQUIRE(ctx, ...) \
    if (any_null({__VA_ARGS__})) return fail(ctx, HP_EINVAL, "NULL pointer argument")
// the kernels move 16 bytes per lane: device operands must be 16-byte aligned (every allocator's blocks are)
inline bool any_misaligned(std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if ((uintptr_t)p & 15u) return true;
    return false;
}
#define HP_ALIGNED(ctx, ...) \
    if (any_misaligned({__VA_ARGS__})) return fail(ctx, HP_EINVAL, "device pointers must be 16-byte aligned")

struct Guard {
    hp_ctx *ctx;
    std::unique_lock<std::mutex> lk;
    explicit Guard(hp_ctx *c) : ctx(c), lk(c->mu) { (void)hipSetDevice(c->device); }
};

int upload(hp_ctx *ctx, const void *host, size_t bytes, void **dptr) {
    HIP_TRY(ctx, hipMalloc(dptr, bytes));
    HIP_TRY(ctx, hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice));
    return HP_OK;
}

// twiddle tables of one (modulus, logn), built on first use (the reference fills global maps
// lazily in the same way, ntt.cpp:117-143)
int get_tables_impl(hp_ctx *ctx, u64 q, size_t logn, DevTables &out) {
    auto key = std::make_pair(q, logn);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        out = it->second;
        return HP_OK;
    }
    std::string why = hp::check_ntt_modulus(q, logn);
    if (!why.empty()) return fail(ctx, HP_EINVAL, why);
    std::vector<hp::Pair> fwd, inv, fk, ik;
    hp::build_fwd_ref(q, logn, fwd);
    hp::build_inv_ref(q, logn, inv);
    DevTables t;
    int rc;
    if ((rc = upload(ctx, fwd.data(), fwd.size() * sizeof(hp::Pair), (void **)&t.fwd_ref))) return rc;
    if ((rc = upload(ctx, inv.data(), inv.size() * sizeof(hp::Pair), (void **)&t.inv_ref))) return rc;
    if (logn >= 11 && logn <= 15) {
        hp::build_fwd_fast(fwd, logn, fk);
        hp::build_inv_fast(inv, logn, ik);
        if ((rc = upload(ctx, fk.data(), fk.size() * sizeof(hp::Pair), (void **)&t.fwd_k))) return rc;
        if ((rc = upload(ctx, ik.data(), ik.size() * sizeof(hp::Pair), (void **)&t.inv_k))) return rc;
    }
    ctx->tables[key] = t;
    out = t;
    return HP_OK;
}
int get_tables(hp_ctx *ctx, u64 q, size_t logn, DevTables &out) {
    return contained(ctx, [&] { return get_tables_impl(ctx, q, logn, out); });
}

// device array of per-limb constants for a modulus chain; with_ntt == false skips the twiddles
int get_plan_impl(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, bool with_ntt, const Plan **out) {
    if (count == 0 || count > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "unsupported number of RNS components");
    std::vector<u64> mv(moduli, moduli + count);
    for (u64 q : mv)
        if (q < 2) return fail(ctx, HP_EINVAL, "modulus must be >= 2");
    auto key = std::make_pair(with_ntt ? logn : (size_t)0, mv);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) {
        *out = &it->second;
        return HP_OK;
    }
    Plan plan;
    std::vector<HpLimb> limbs(count);
    for (size_t k = 0; k < count; k++) {
        hp::ModConsts c = hp::make_consts(mv[k]);
        plan.consts.push_back(c);
        HpLimb &l = limbs[k];
        memset(&l, 0, sizeof(l));
        l.q = c.q; l.two_q = c.two_q; l.neg_q = c.neg_q; l.mqinv = c.mqinv; l.r64 = c.r64; l.r64h = c.r64h;
        l.barrett_c = c.barrett_c; l.k = c.k; l.fix = c.fix;
        if (with_ntt) {
            DevTables t;
            int rc = get_tables(ctx, mv[k], logn, t);
            if (rc) return rc;
            l.fwd_ref = t.fwd_ref; l.inv_ref = t.inv_ref; l.fwd_k = t.fwd_k; l.inv_k = t.inv_k;
        }
    }
    int rc = upload(ctx, limbs.data(), limbs.size() * sizeof(HpLimb), (void **)&plan.d_limbs);
    if (rc) return rc;
    auto ins = ctx->plans.emplace(key, std::move(plan));
    *out = &ins.first->second;
    return HP_OK;
}
int get_plan(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, bool with_ntt, const Plan **out) {
    return contained(ctx, [&] { return get_plan_impl(ctx, logn, moduli, count, with_ntt, out); });
}

// grow-only workspace; stream order makes reuse across calls safe
int ws_reserve(hp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return HP_OK;
    if (ctx->ws) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) {
        ctx->err = std::string("workspace hipMalloc: ") + hipGetErrorString(e);
        return HP_ENOMEM;
    }
    ctx->ws_bytes = bytes;
    return HP_OK;
}

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    u64 *take(size_t words) {
        u64 *p = (u64 *)(base + off);
        off += (words * 8 + 255) & ~(size_t)255;
        return p;
    }
};
inline size_t padded(size_t words) { return (words * 8 + 255) & ~(size_t)255; }

// ---- profiling brackets ---------------------------------------------------------
hipEvent_t get_event(hp_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    hp_ctx *ctx;
    bool on;
    ProfEvent ev;
    ProfScope(hp_ctx *c, const char *family) : ctx(c) {
        on = c->prof_on && c->prof_family == family;
        if (on) {
            ev.a = get_event(c);
            ev.b = get_event(c);
            (void)hipEventRecord(ev.a, c->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(ev.b, ctx->stream);
            ctx->prof_events.push_back(ev);
        }
    }
};

// ---- kernel-launch wrappers --------------------------------------------------------
int run_ntt(hp_ctx *ctx, const HpNttJob &job) {
    if (job.W == 0) return HP_OK;
    hipError_t e;
    {
        ProfScope ps(ctx, job.inverse ? "intt" : "ntt");
        if (!ctx->force_generic && job.logn >= 11 && job.logn <= 15) e = hp_launch_ntt_fast(job, ctx->stream);
        else e = hp_launch_ntt_generic(job, ctx->stream);
    }
    if (e != hipSuccess) return fail(ctx, HP_EHIP, std::string("transform launch: ") + hipGetErrorString(e));
    return HP_OK;
}

int chk(hp_ctx *ctx, hipError_t e, const char *what) {
    if (e != hipSuccess) return fail(ctx, HP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return HP_OK;
}

HpNttJob batch_job(const Plan *plan, size_t logn, size_t L, size_t P, const u64 *src, u64 *dst, size_t src_ps,
                   size_t dst_ps, int inverse, int strict) {
    HpNttJob j;
    memset(&j, 0, sizeof(j));
    j.limbs = plan->d_limbs; j.src = src; j.dst = dst; j.logn = (u32)logn; j.L = (u32)L; j.P = (u32)P;
    j.src_pstride = (u32)src_ps; j.dst_pstride = (u32)dst_ps; j.src_kstride = 1; j.W = (u32)(L * P); j.mode = HP_NTT_BATCH;
    j.inverse = inverse; j.strict = strict;
    return j;
}

// gather map of cycle(poly, step) (permutation.cpp:39-53): out[to] = in[perm[to]]; cached per (logn, step)
int get_cycle_perm_impl(hp_ctx *ctx, size_t logn, size_t step, const u32 **out) {
    const size_t n = (size_t)1 << logn;
    auto key = std::make_pair(logn, step);
    auto it = ctx->perms.find(key);
    if (it == ctx->perms.end()) {
        std::vector<u32> perm(n);
        for (size_t i = 0; i < n; i++) perm[i] = (u32)i;
        const u32 mask = (u32)((1u << (logn + 1)) - 1);
        u32 factor = 1;
        for (size_t s = 0; s < step; s++) factor *= 3u;
        factor &= mask;
        u32 pw = 1;
        for (size_t i = 0; i < n / 2; i++, pw *= 3u) {
            const u32 old_idx = pw & mask;
            const u32 from = hp::bit_rev((old_idx - 1) / 2, (int)logn);
            const u32 to = hp::bit_rev((((old_idx * factor) & mask) - 1) / 2, (int)logn);
            perm[to] = from;
            perm[n - 1 - to] = (u32)(n - 1 - from);
        }
        u32 *d = nullptr;
        int rc = upload(ctx, perm.data(), n * sizeof(u32), (void **)&d);
        if (rc) return rc;
        it = ctx->perms.emplace(key, d).first;
    }
    *out = it->second;
    return HP_OK;
}
int get_cycle_perm(hp_ctx *ctx, size_t logn, size_t step, const u32 **out) {
    return contained(ctx, [&] { return get_cycle_perm_impl(ctx, logn, step, out); });
}

bool logn_ok(size_t logn) { return logn >= 1 && logn <= 15; }

// rgsw.cpp:57-156 on a batch.  pt rows: polynomial p at pt + p*pt_pstride limbs.
// workspace: coef [P][L][N], digits [P][L][L+1][N]
size_t ext_prod_ws_words(size_t n, size_t L, size_t P) { return padded(P * L * n) / 8 + padded(P * L * (L + 1) * n) / 8; }

// (i) c[j] = strict(INTT(pt[j])) for the digits j in [j0, j1)                          rgsw.cpp:103-105
int ks_coef(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t j0, size_t j1, const u64 *pt,
            size_t pt_pstride, u64 *coef) {
    const size_t n = (size_t)1 << logn;
    HpNttJob j = batch_job(plan, logn, j1 - j0, P, pt + j0 * n, coef + j0 * n, pt_pstride, L, 1, 1);
    j.limbs = plan->d_limbs + j0;
    return run_ntt(ctx, j);
}

// (ii) + (iii) for the output moduli k in [k0, k1) of q_0..q_{L-1}, p: every digit limb is needed, only the
// owned columns of digits / key / out are touched
// key_L0: number of ciphertext moduli the key was generated for (>= L; its polynomials have key_L0 + 1 limbs)
int ks_digits_inner(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t k0, size_t k1, const u64 *coef,
                    const u64 *pt, size_t pt_pstride, const u64 *key, size_t key_L0, u64 *out, u64 *digits) {
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
    if ((rc = run_ntt(ctx, sj))) return rc;
    // (iii) u128 inner product + Montgomery                         rgsw.cpp:121-153
    {
        ProfScope ps(ctx, "ks_inner");
        rc = chk(ctx, hp_launch_ks_inner(plan->d_limbs, (u32)L, (u32)k0, (u32)(k1 - k0), (u32)(key_L0 + 1), (u32)n, (u32)P, digits, pt,
                                         (u32)pt_pstride, key, out, ctx->stream), "ks_inner");
    }
    return rc;
}

int ext_prod(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, const u64 *pt, size_t pt_pstride,
             const u64 *key, size_t key_L0, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn;
    u64 *coef = cv.take(P * L * n);
    u64 *digits = cv.take(P * L * (L + 1) * n);
    int rc;
    if ((rc = ks_coef(ctx, plan, logn, L, P, 0, L, pt, pt_pstride, coef))) return rc;
    return ks_digits_inner(ctx, plan, logn, L, P, 0, L + 1, coef, pt, pt_pstride, key, key_L0, out, digits);
}

// rescaling.cpp:46-75 / mod_switch.cpp:45-77 on P2 polynomials of L limbs (x rows: poly p2 at x + p2*L limbs)
size_t drop_ws_words(size_t n, size_t L, size_t P2) { return padded(P2 * n) / 8 + padded(P2 * (L - 1) * n) / 8; }

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
    if (bgv) {
        const u64 s = hp::inverse_mod_prime(t, q_last) % q_last;
        lj.post_scalar = s;
        lj.post_scalar_h = hp::harvey_quotient(s, q_last);
        lj.use_post_scalar = 1;
    }
    return run_ntt(ctx, lj);
}

// out[k] = ((x[k] - NTT_k(centre(barrett_k(clast)))) * inv_k) [* (q_last mod t)] [+ addend[k]] for the limbs k in [k0, k1)
// of the L-1 that remain.  rem: workspace of P2*(k1-k0)*n words (unused by the fused tiled path).
int drop_apply(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, size_t k0, size_t k1, const HpDropConsts &dc0,
               const u64 *x, const u64 *clast, const u64 *addend, size_t add_poly_stride, size_t add_ct_stride, u32 add_mask,
               u64 *out, u64 *rem) {
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
    if (!ctx->force_generic && logn >= 11 && logn <= 15 && !ctx->no_fused_drop) {
        HpNttJob fj = batch_job(plan, logn, kc, P2, clast, nullptr, 1, 0, 0, 0);
        fj.limbs = limbs;
        fj.src_kstride = 0;
        fj.pair_moduli = (u32)ctx->drop_group;
        if (fj.pair_moduli > kc) fj.pair_moduli = (u32)kc;
        HpDropArgs da;
        memset(&da, 0, sizeof(da));
        da.dc = dc; da.x = x; da.L = (u32)L; da.addend = addend; da.add_poly_stride = (u32)add_poly_stride;
        da.add_ct_stride = (u32)add_ct_stride; da.add_mask = add_mask; da.out = out; da.out_stride = (u32)(L - 1);
        ProfScope ps(ctx, "ntt_drop");   // its own family: a different kernel (k_ntt_fwd_drop) with 2-3x the bytes of a plain transform
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "fused drop NTT");
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

int drop_last(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, bool bgv, u64 t, const u64 *x,
              const u64 *addend, size_t add_poly_stride, size_t add_ct_stride, u32 add_mask, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn;
    HpDropConsts dc;
    make_drop_consts(plan, L, bgv, t, dc);
    u64 *clast = cv.take(P2 * n);
    u64 *rem = cv.take(P2 * (L - 1) * n);
    int rc;
    if ((rc = drop_coeffs(ctx, plan, logn, L, P2, bgv, t, x, clast))) return rc;
    return drop_apply(ctx, plan, logn, L, P2, 0, L - 1, dc, x, clast, addend, add_poly_stride, add_ct_stride, add_mask, out, rem);
}

int check_ext_args(hp_ctx *ctx, size_t logn, size_t L, size_t batch) {
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (L < 1 || L + 1 > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "Invalid component number in RGSW ciphertext.");
    if (batch == 0) return fail(ctx, HP_EINVAL, "empty batch");
    return HP_OK;
}

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
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char *hp_version(void) { return "hehub_amd 0.1 (gfx950)"; }

int hp_ctx_create(int device, hp_ctx **out) {
    if (!out) return HP_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return HP_EHIP;
    if (hipSetDevice(device) != hipSuccess) return HP_EHIP;
    hp_ctx *c = new (std::nothrow) hp_ctx();
    if (!c) return HP_ENOMEM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return HP_EHIP;
    }
    c->stream = c->own_stream;
    c->no_fused_drop = getenv("HP_NO_FUSED_DROP") != nullptr;
    c->hks_two_step = getenv("HP_HKS_TWO_STEP") != nullptr;
    if (const char *e = getenv("HP_DROP_GROUP")) c->drop_group = atoi(e) > 0 ? atoi(e) : 0;
    if (const char *e = getenv("HP_SPREAD_GROUP")) c->spread_group = atoi(e) > 0 ? atoi(e) : 0;   // measured: 2..6 alike, -2 % on the launch
    if (const char *e = getenv("HP_MULT_STREAMS")) c->mult_streams = atoi(e) >= 2 ? 2 : 1;
    if (const char *e = getenv("HP_MULT_CHUNK")) c->mult_chunk = (size_t)atol(e);
    *out = c;
    return HP_OK;
}

void hp_ctx_destroy(hp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : ctx->tables) {
        (void)hipFree(kv.second.fwd_ref); (void)hipFree(kv.second.inv_ref);
        (void)hipFree(kv.second.fwd_k); (void)hipFree(kv.second.inv_k);
    }
    for (auto &kv : ctx->plans) (void)hipFree(kv.second.d_limbs);
    for (auto &kv : ctx->perms) (void)hipFree(kv.second);
    for (auto &kv : ctx->crt) (void)hipFree(kv.second);
    for (auto &kv : ctx->hks) (void)hipFree(kv.second);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) {
        if (ctx->aux[i]) (void)hipStreamDestroy(ctx->aux[i]);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *hp_last_error(hp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int hp_ctx_set_stream(hp_ctx *ctx, void *s) {
    Guard g(ctx);
    ctx->stream = (hipStream_t)s;   // NULL is the HIP default (null) stream, e.g. torch's default stream
    return HP_OK;
}
int hp_ctx_reset_stream(hp_ctx *ctx) {
    Guard g(ctx);
    ctx->stream = ctx->own_stream;
    return HP_OK;
}
void *hp_ctx_get_stream(hp_ctx *ctx) { return (void *)ctx->stream; }

// The scratch workspace only grows (the largest call so far defines it: 10+ GiB for a C3 batch of 256); this gives it back.
int hp_ctx_release_workspace(hp_ctx *ctx) {
    Guard g(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) HIP_TRY(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    return HP_OK;
}
size_t hp_ctx_workspace_bytes(hp_ctx *ctx) { return ctx ? ctx->ws_bytes : 0; }

int hp_sync(hp_ctx *ctx) {
    Guard g(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}

int hp_dev_alloc(hp_ctx *ctx, size_t bytes, void **dptr) {
    Guard g(ctx);
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return fail(ctx, HP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return HP_OK;
}
int hp_dev_free(hp_ctx *ctx, void *dptr) {
    Guard g(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(dptr));
    return HP_OK;
}
int hp_memcpy_h2d(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    Guard g(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}
int hp_memcpy_d2h(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    Guard g(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}
int hp_ctx_set_force_generic(hp_ctx *ctx, int on) {
    Guard g(ctx);
    ctx->force_generic = on != 0;
    return HP_OK;
}

// ---- drop-in, host pointers -----------------------------------------------------------
int hp_ntt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t logn, uint64_t q, uint64_t *x) {
    Guard g(ctx);
    return host_transform(ctx, logn, q, x, 0);
}
int hp_intt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t logn, uint64_t q, uint64_t *x) {
    Guard g(ctx);
    return host_transform(ctx, logn, q, x, 1);
}
int hp_cache_ntt_factors_strict(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count) {
    Guard g(ctx);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    for (size_t i = 0; i < count; i++) {
        DevTables t;
        int rc = get_tables(ctx, moduli[i], logn, t);
        if (rc) return rc;
    }
    return HP_OK;
}
int hp_batched_barrett_lazy(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_BARRETT_LAZY, q, n, v, nullptr, v, 1);
}
int hp_batched_barrett(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_BARRETT, q, n, v, nullptr, v, 1);
}
int hp_batched_reduce_strict(hp_ctx *ctx, uint64_t q, size_t n, uint64_t *v) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_STRICT, q, n, v, nullptr, v, 1);
}
int hp_batched_mul_mod_hybrid_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *a, const uint64_t *b,
                                   uint64_t *out) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_MUL_HYBRID, q, n, a, b, out, 1);
}
int hp_batched_mul_mod_barrett_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *a, const uint64_t *b,
                                    uint64_t *out) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_MUL_BARRETT, q, n, a, b, out, 1);
}
int hp_batched_montgomery_128_lazy(hp_ctx *ctx, uint64_t q, size_t n, const uint64_t *in128, uint64_t *out) {
    Guard g(ctx);
    return host_vec(ctx, HP_V_MONTGOMERY128, q, n, in128, nullptr, out, 2);
}

// ---- device-resident batches ---------------------------------------------------------------
int hp_dev_ntt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, d_x);
    HP_ALIGNED(ctx, d_x);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    return run_ntt(ctx, batch_job(plan, logn, L, batch, d_x, d_x, L, L, 0, 0));
}

int hp_dev_intt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x, int strict) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, d_x);
    HP_ALIGNED(ctx, d_x);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    return run_ntt(ctx, batch_job(plan, logn, L, batch, d_x, d_x, L, L, 1, strict));
}

static int dev_binary(hp_ctx *ctx, int op, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                      const uint64_t *a, const uint64_t *b, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, a, b, out);
    HP_ALIGNED(ctx, a, b, out);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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

int hp_dev_poly_scalar_mul(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                           const uint64_t *rns_scalar, const uint64_t *a, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, rns_scalar, a, out);
    HP_ALIGNED(ctx, a, out);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, x);
    HP_ALIGNED(ctx, x);
    if (n == 0 || (n & (n - 1))) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, 0, moduli, L, false, &plan);
    if (rc) return rc;
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_poly_strict(plan->d_limbs, (u32)L, (u32)n, (u32)(batch * L), x, ctx->stream), "poly_strict");
}

int hp_dev_poly_involution(hp_ctx *ctx, size_t logn, size_t L, size_t batch, const uint64_t *in, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, in, out);
    HP_ALIGNED(ctx, in, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (batch == 0) return HP_OK;
    if (in == out) return fail(ctx, HP_EINVAL, "involution cannot run in place");
    ProfScope ps(ctx, "elem");
    return chk(ctx, hp_launch_reverse((u32)1 << logn, (u32)(batch * L), in, out, ctx->stream), "involution");
}

int hp_dev_poly_cycle(hp_ctx *ctx, size_t logn, size_t L, size_t batch, size_t step, const uint64_t *in,
                      uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, in, out);
    HP_ALIGNED(ctx, in, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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

int hp_dev_mult_low_level(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                          const uint64_t *ct1, const uint64_t *ct2, uint64_t *quad) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, ct1, ct2, quad);
    HP_ALIGNED(ctx, ct1, ct2, quad);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, key, out);
    HP_ALIGNED(ctx, pt, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
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

static int dev_drop(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, bool bgv, uint64_t t, size_t batch,
                    const uint64_t *ct, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, ct, out);
    HP_ALIGNED(ctx, ct, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (bgv && t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    if (batch == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    const size_t n = (size_t)1 << logn;
    if ((rc = ws_reserve(ctx, drop_ws_words(n, L, 2 * batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return drop_last(ctx, plan, logn, L, 2 * batch, bgv, t, ct, nullptr, 0, 0, 0, out, cv);
}
int hp_dev_ckks_rescale(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, const uint64_t *ct,
                        uint64_t *out) { return dev_drop(ctx, logn, L, moduli, false, 0, batch, ct, out); }
// extension (the reference throws "under development" for dropping_primes >= 2, rescaling.cpp:83-85): `drops` successive
// exact one-prime drops; tmp holds the intermediate levels in two alternating halves of batch*2*(L-1)*N words each
// (unused when drops == 1)
int hp_dev_ckks_rescale_n(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t drops, size_t batch,
                          const uint64_t *ct, uint64_t *tmp, uint64_t *out) {
    if (drops < 1 || drops >= L) return fail(ctx, HP_EINVAL, "The number of primes to be dropped is not positive.");
    if (drops > 1 && !tmp) return fail(ctx, HP_EINVAL, "rescale by several primes needs the intermediate buffer");
    const size_t n = (size_t)1 << logn, half = batch * 2 * (L - 1) * n;
    const uint64_t *src = ct;
    for (size_t d = 0; d < drops; d++) {
        uint64_t *dst = (d + 1 == drops) ? out : tmp + (d & 1) * half;
        int rc = dev_drop(ctx, logn, L - d, moduli, false, 0, batch, src, dst);
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
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, quad, key, out);
    HP_ALIGNED(ctx, quad, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (bgv && inner_t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
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
static int dev_ckks_automorphism(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                 bool conj, size_t step, const uint64_t *ct, const uint64_t *key, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, ct, key, out);
    HP_ALIGNED(ctx, ct, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (!conj && step >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    const size_t n = (size_t)1 << logn;
    const size_t words = padded(batch * 2 * L * n) / 8 + padded(batch * 2 * (L + 1) * n) / 8 + ext_prod_ws_words(n, L, batch) +
                         drop_ws_words(n, L + 1, 2 * batch);
    if ((rc = ws_reserve(ctx, words * 8))) return rc;
    Carver cv(ctx->ws);
    u64 *moved = cv.take(batch * 2 * L * n);
    u64 *ext = cv.take(batch * 2 * (L + 1) * n);
    {
        ProfScope ps(ctx, "elem");
        if (conj) {
            rc = chk(ctx, hp_launch_reverse((u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "involution");
        } else {
            const u32 *perm;
            if ((rc = get_cycle_perm(ctx, logn, step, &perm))) return rc;
            rc = chk(ctx, hp_launch_gather(perm, (u32)n, (u32)(batch * 2 * L), ct, moved, ctx->stream), "cycle");
        }
    }
    if (rc) return rc;
    if ((rc = ext_prod(ctx, plan, logn, L, batch, moved + L * n, 2 * L, key, key_L0, ext, cv))) return rc;
    return drop_last(ctx, plan, logn, L + 1, 2 * batch, false, 0, ext, moved, L, 2 * L, 1, out, cv);
}

// mult_low_level + relinearize + drop q_last, processed in sub-batches so the working set
// (dominated by the L(L+1) digit limbs per ciphertext) stays small
static int dev_mult(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, bool bgv, u64 t, u64 inner_t, size_t batch,
                    const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, ct1, ct2, key, out);
    HP_ALIGNED(ctx, ct1, ct2, key, out);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = key_level_ok(ctx, L, key_L0))) return rc;
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (bgv && t == 0) return fail(ctx, HP_EINVAL, "plain modulus must be positive");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
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
        {
            ProfScope ps(ctx, "tensor");
            rc = chk(ctx, hp_launch_tensor(plan->d_limbs, (u32)L, 0, (u32)L, (u32)n, (u32)P, ct1 + b0 * 2 * L * n,
                                           ct2 + b0 * 2 * L * n, quad, ctx->stream), "tensor");
        }
        // the reference's bgv::relinearize runs its inner mod switch with plain_modulus == 1 (bgv.h:32): inner_t = 1
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

// ---- profiling ------------------------------------------------------------------------------
// ---- either side of the path: encrypt / decrypt cores, RNS base transforms ------------------------
int hp_dev_rlwe_encrypt_core(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, const int64_t *noise,
                             const uint64_t *c1, const uint64_t *pt, const uint64_t *sk, uint64_t *ct) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, noise, c1, pt, sk, ct);
    HP_ALIGNED(ctx, noise, c1, pt, sk, ct);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, ct, sk, pt);
    HP_ALIGNED(ctx, ct, sk, pt);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    Guard g(ctx);
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
    Guard g(ctx);
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

// constants of the CRT branch of the many -> one base transform (rns_transform.cpp:86-104), cached per (moduli, t)
static int get_crt_consts(hp_ctx *ctx, const uint64_t *moduli, size_t L, u64 t, const HpCrtConsts **out) {
    return contained(ctx, [&] {
        auto key = std::make_pair(std::vector<u64>(moduli, moduli + L), t);
        auto it = ctx->crt.find(key);
        if (it == ctx->crt.end()) {
            HpCrtConsts c;
            memset(&c, 0, sizeof(c));
            c.t = t;
            typedef unsigned __int128 u128;
            u64 prod_t = 1 % t;
            for (size_t a = 0; a < L; a++) {
                c.pref[a] = prod_t;
                c.pref_h[a] = hp::harvey_quotient(prod_t, t);
                prod_t = (u64)((u128)prod_t * (moduli[a] % t) % t);
                for (size_t b = 0; b < a; b++) {
                    if (moduli[b] % moduli[a] == 0) return fail(ctx, HP_EINVAL, "moduli are not pairwise coprime");
                    c.inv[b][a] = hp::inverse_mod_prime(moduli[b] % moduli[a], moduli[a]) % moduli[a];
                    c.inv_h[b][a] = hp::harvey_quotient(c.inv[b][a], moduli[a]);
                }
            }
            c.q_mod_t = prod_t;
            // floor(Q/2) = (Q-1)/2 has the residues (q_a - 1)/2; its mixed-radix digits by the same recurrence
            for (size_t a = 0; a < L; a++) {
                const u64 qa = moduli[a];
                u64 u = (qa - 1) / 2;
                for (size_t b = 0; b < a; b++) {
                    const u64 vb = c.half[b] % qa;
                    u = (u64)((u128)((u + qa - vb) % qa) * c.inv[b][a] % qa);
                }
                c.half[a] = u;
            }
            HpCrtConsts *d = nullptr;
            int rc = upload(ctx, &c, sizeof(c), (void **)&d);
            if (rc) return rc;
            it = ctx->crt.emplace(key, d).first;
        }
        *out = it->second;
        return (int)HP_OK;
    });
}

// rns_transform.h rns_base_transform(poly, {new_modulus}) complete: small-coefficient branch for the polynomials whose
// coefficients are all small, CRT composition for the others -- decided per polynomial on the device, no host round trip
int hp_dev_rns_base_to_single(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, uint64_t new_modulus, size_t batch,
                              const uint64_t *in, uint64_t *out) {
    Guard g(ctx);
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
    Guard g(ctx);
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

// ---- hybrid key switch (extension; kernels in hp_hks.hip) -------------------------------------------
static int get_hks_consts(hp_ctx *ctx, const uint64_t *mext, size_t L, size_t k, size_t alpha, const HpHksConsts **out) {
    return contained(ctx, [&] {
        const size_t E = L + k, nd = (L + alpha - 1) / alpha;
        auto key = std::make_pair(std::vector<u64>(mext, mext + E), std::make_pair(k, alpha));
        auto it = ctx->hks.find(key);
        if (it == ctx->hks.end()) {
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
            if (k <= HP_HKS_MAX_ALPHA) {
                const uint64_t *pm = mext + L;
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
                    c.p_mod_q[i] = prod;
                    c.p_mod_q_h[i] = hp::harvey_quotient(prod, mext[i]);
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

static bool fused_drop_ok(const hp_ctx *ctx, size_t logn) { return !ctx->force_generic && logn >= 11 && logn <= 15 && !ctx->no_fused_drop; }

static int hks_switch(hp_ctx *ctx, const Plan *plan, const HpHksConsts *hc, size_t logn, size_t L, size_t k, size_t alpha, size_t P,
                      const u64 *pt, size_t pt_pstride, const u64 *key, const u64 *addend, size_t add_poly_stride,
                      size_t add_ct_stride, u32 add_mask, const uint64_t *mext, u64 *out, Carver &cv) {
    const size_t n = (size_t)1 << logn, E = L + k;
    u64 *ks, *rem;
    int rc;
    if ((rc = hks_front(ctx, plan, hc, logn, L, k, alpha, P, pt, pt_pstride, key, mext, &ks, &rem, cv))) return rc;
    // transform of the remainders with the rest of ModDown fused into its stores: out = (x - NTT(rem)) * P^-1 [+ addend]
    if (!ctx->force_generic && logn >= 11 && logn <= 15 && !ctx->no_fused_drop) {
        HpNttJob fj = batch_job(plan, logn, L, 2 * P, rem, nullptr, L, 0, 0, 0);
        HpDropArgs da;
        memset(&da, 0, sizeof(da));
        da.raw_input = 1;
        for (size_t i = 0; i < L; i++) { da.dc.inv[i] = hc_host_pinv(ctx, mext, L, k, i, &da.dc.inv_h[i]); }
        da.x = ks; da.L = (u32)E; da.addend = addend; da.add_poly_stride = (u32)add_poly_stride; da.add_ct_stride = (u32)add_ct_stride;
        da.add_mask = addend ? add_mask : 0u; da.out = out; da.out_stride = (u32)L;
        ProfScope ps(ctx, "ntt_drop");
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks fused ModDown");
    }
    if ((rc = run_ntt(ctx, batch_job(plan, logn, L, 2 * P, rem, rem, L, L, 0, 0)))) return rc;
    ProfScope ps(ctx, "hks_down_fin");
    return chk(ctx, hp_launch_hks_down_fin(plan->d_limbs, hc, (u32)L, (u32)n, (u32)(2 * P), ks, rem, addend, (u32)add_poly_stride,
                                           (u32)add_ct_stride, add_mask, out, ctx->stream), "hks_down_fin");
}

static int hks_args_ok(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, size_t batch) {
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (L < 1 || k < 1 || k > HP_CRT_MAX_LIMBS || L + k > HP_MAX_LIMBS) return fail(ctx, HP_EINVAL, "unsupported number of moduli");
    if (alpha < 1 || alpha > HP_HKS_MAX_ALPHA || (L + alpha - 1) / alpha > HP_HKS_MAX_DIGITS)
        return fail(ctx, HP_EINVAL, "unsupported digit size");
    if (batch == 0) return fail(ctx, HP_EINVAL, "empty batch");
    return HP_OK;
}

int hp_dev_hks_switch(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                      const uint64_t *pt, const uint64_t *key, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, key, out);
    HP_ALIGNED(ctx, pt, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, moduli_ext, L, k, alpha, &hc))) return rc;
    const size_t n = (size_t)1 << logn, nd = (L + alpha - 1) / alpha;
    if ((rc = ws_reserve(ctx, hks_ws_words(n, L, k, nd, batch) * 8))) return rc;
    Carver cv(ctx->ws);
    return hks_switch(ctx, plan, hc, logn, L, k, alpha, batch, pt, L, key, nullptr, 0, 0, 0, moduli_ext, out, cv);
}

// ckks rotate / conjugate with a hybrid key: moved = gather(ct); out = hks_switch(moved[1]); out[0] += moved[0]
static int dev_hks_automorphism(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *mext, size_t batch,
                                bool conj, size_t step, const uint64_t *ct, const uint64_t *key, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, mext, ct, key, out);
    HP_ALIGNED(ctx, ct, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    if (!conj && step >= ((size_t)1 << 17)) return fail(ctx, HP_EINVAL, "rotation step out of range");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, mext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, mext, L, k, alpha, &hc))) return rc;
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
int hp_dev_ckks_rotate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                           size_t step, const uint64_t *ct, const uint64_t *rot_key, uint64_t *out) {
    return dev_hks_automorphism(ctx, logn, L, k, alpha, moduli_ext, batch, false, step, ct, rot_key, out);
}
int hp_dev_ckks_conjugate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                              const uint64_t *ct, const uint64_t *conj_key, uint64_t *out) {
    return dev_hks_automorphism(ctx, logn, L, k, alpha, moduli_ext, batch, true, 0, ct, conj_key, out);
}

// ckks::mult_low_level + relinearisation with a hybrid key + rescale by the last ciphertext modulus
int hp_dev_ckks_mult_relin_rescale_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext,
                                       size_t batch, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *key,
                                       uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, ct1, ct2, key, out);
    HP_ALIGNED(ctx, ct1, ct2, key, out);
    int rc = hks_args_ok(ctx, logn, L, k, alpha, batch);
    if (rc) return rc;
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + k, true, &plan))) return rc;
    const HpHksConsts *hc;
    if ((rc = get_hks_consts(ctx, moduli_ext, L, k, alpha, &hc))) return rc;
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
            if ((rc = chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks ModDown of the last limb"))) return rc;
        }
        {
            HpNttJob lj = batch_job(plan, logn, 1, P2, r_last, c_last, 1, 1, 1, 1);
            lj.limbs = plan->d_limbs + (L - 1);
            if ((rc = run_ntt(ctx, lj))) return rc;
        }
        const bool in_loads = getenv("HP_HKS_COMBINE_KERNEL") == nullptr;   // tuning switch: the combination as its own kernel
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
        return chk(ctx, hp_launch_ntt_fast_drop(fj, da, ctx->stream), "hks fused ModDown + rescale");
    }
    if ((rc = hks_switch(ctx, plan, hc, logn, L, k, alpha, batch, quad + 2 * L * n, 3 * L, key, quad, L, 3 * L, 3, moduli_ext, lin, cv)))
        return rc;
    return drop_last(ctx, plan, logn, L, 2 * batch, false, 0, lin, nullptr, 0, 0, 0, out, cv);
}

// ---- limb-range stages (limb-sharded "latency" mode across GPUs) ---------------------------------
static int range_ok(hp_ctx *ctx, size_t lo, size_t hi, size_t limit) {
    if (lo > hi || hi > limit) return fail(ctx, HP_EINVAL, "limb range out of bounds");
    return HP_OK;
}

int hp_dev_mult_low_level_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, size_t k0,
                                size_t k1, const uint64_t *ct1, const uint64_t *ct2, uint64_t *quad) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, ct1, ct2, quad);
    HP_ALIGNED(ctx, ct1, ct2, quad);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli_ext, pt, coef);
    HP_ALIGNED(ctx, pt, coef);
    int rc = check_ext_args(ctx, logn, L, batch);
    if (rc || (rc = range_ok(ctx, j0, j1, L))) return rc;
    if (j0 == j1) return HP_OK;
    const Plan *plan;
    if ((rc = get_plan(ctx, logn, moduli_ext, L + 1, true, &plan))) return rc;
    return ks_coef(ctx, plan, logn, L, batch, j0, j1, pt, pt_pstride, coef);
}

int hp_dev_ks_inner_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0, size_t k1,
                          const uint64_t *coef, const uint64_t *pt, size_t pt_pstride, const uint64_t *key, uint64_t *out) {
    Guard g(ctx);
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
    return ks_digits_inner(ctx, plan, logn, L, batch, k0, k1, coef, pt, pt_pstride, key, L, out, digits);
}

int hp_dev_drop_coeffs(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                       const uint64_t *x, uint64_t *clast) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, x, clast);
    HP_ALIGNED(ctx, x, clast);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
    if (L < 2) return fail(ctx, HP_EINVAL, "Unable to drop the only one prime.");
    if (P2 == 0) return HP_OK;
    const Plan *plan;
    int rc = get_plan(ctx, logn, moduli, L, true, &plan);
    if (rc) return rc;
    return drop_coeffs(ctx, plan, logn, L, P2, plain_modulus != 0, plain_modulus, x, clast);
}

int hp_dev_drop_apply_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                            size_t k0, size_t k1, const uint64_t *x, const uint64_t *clast, const uint64_t *addend,
                            size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask, uint64_t *out) {
    Guard g(ctx);
    HP_REQUIRE(ctx, moduli, x, clast, out);
    HP_ALIGNED(ctx, x, clast, out);
    if (!logn_ok(logn)) return fail(ctx, HP_EUNSUPPORTED, "ring degrees 2^1 .. 2^15 are supported");
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
    return drop_apply(ctx, plan, logn, L, P2, k0, k1, dc, x, clast, addend, add_poly_stride, add_ct_stride, add_mask, out, rem);
}

int hp_prof_begin(hp_ctx *ctx, const char *family) {
    Guard g(ctx);
    return contained(ctx, [&] {
        for (auto &ev : ctx->prof_events) { ctx->event_pool.push_back(ev.a); ctx->event_pool.push_back(ev.b); }
        ctx->prof_events.clear();
        ctx->prof_family = family ? family : "";
        ctx->prof_on = true;
        return (int)HP_OK;
    });
}
int hp_prof_end(hp_ctx *ctx, size_t *launches, double *total_ms) {
    Guard g(ctx);
    ctx->prof_on = false;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0;
    for (auto &ev : ctx->prof_events) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
        total += ms;
        ctx->event_pool.push_back(ev.a);
        ctx->event_pool.push_back(ev.b);
    }
    if (launches) *launches = ctx->prof_events.size();
    if (total_ms) *total_ms = total;
    ctx->prof_events.clear();
    return HP_OK;
}

} // extern "C"
