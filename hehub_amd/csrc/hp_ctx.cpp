// hp_ctx.cpp -- engine context: life cycle, stream selection, scratch workspace, device memory helpers, and the
// caches of twiddle tables / per-limb constants / gather maps / CRT constants (include/hehub_amd.h, "engine").
#include "hp_ctx.h"

#include <cstdlib>
#include <cstring>

namespace hpi {

namespace {
// last failure of the calling thread: (context, message)
thread_local const hp_ctx *tl_err_ctx = nullptr;
thread_local std::string tl_err_msg;
} // namespace

int fail(hp_ctx *ctx, int code, const std::string &msg) {
    ctx->err = msg;        // under the context lock (every caller holds it)
    tl_err_ctx = ctx;
    tl_err_msg = msg;
    return code;
}

// for code that runs WITHOUT the context lock (the node layer's worker threads between entry points): the message goes to
// the calling thread's slot only, so hp_last_error() on that thread returns it and ctx->err is not written unlocked
int fail_local(hp_ctx *ctx, int code, const std::string &msg) {
    tl_err_ctx = ctx;
    tl_err_msg = msg;
    return code;
}
int chk_local(hp_ctx *ctx, hipError_t e, const char *what) {
    if (e != hipSuccess) return fail_local(ctx, HP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return HP_OK;
}

int chk(hp_ctx *ctx, hipError_t e, const char *what) {
    if (e != hipSuccess) return fail(ctx, HP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return HP_OK;
}

int upload(hp_ctx *ctx, const void *host, size_t bytes, void **dptr) {
    HIP_TRY(ctx, hipMalloc(dptr, bytes));
    HIP_TRY(ctx, hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice));
    return HP_OK;
}

// twiddle tables of one (modulus, logn), built on first use (the reference fills global maps
// lazily in the same way, ntt.cpp:117-143)
static int get_tables_impl(hp_ctx *ctx, u64 q, size_t logn, DevTables &out) {
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

static int get_plan_impl(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, bool with_ntt, const Plan **out) {
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
    plan.logn = with_ntt ? logn : 0;
    auto ins = ctx->plans.emplace(key, std::move(plan));
    *out = &ins.first->second;
    return HP_OK;
}
int get_plan(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, bool with_ntt, const Plan **out) {
    return contained(ctx, [&] { return get_plan_impl(ctx, logn, moduli, count, with_ntt, out); });
}

// Parity level A: per-limb records + tables of doubles for a plan whose moduli are all below 2^50 and whose ring degree has
// tiled kernels.  Built from the same host tables as level B (pairs_to_f64), cached per (q, logn).
static int ensure_plan_a_impl(hp_ctx *ctx, const Plan *plan, bool *ok) {
    *ok = false;
    if (plan->a_state == 1) { *ok = true; return HP_OK; }
    if (plan->a_state < 0) return HP_OK;
    const size_t logn = plan->logn;
    bool can = logn >= 11 && logn <= 15;   // (the debug switches that bypass the tiled kernels are looked at per call: LevelScope)
    for (const hp::ModConsts &c : plan->consts)
        if (c.q >= ((u64)1 << 50) || c.q < 3) can = false;
    if (!can) { plan->a_state = -1; return HP_OK; }
    if (!ctx->sh->range_flag) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->sh->range_flag, sizeof(u32)));
        HIP_TRY(ctx, hipMemset(ctx->sh->range_flag, 0, sizeof(u32)));
    }
    std::vector<HpLimbA> limbs(plan->consts.size());
    for (size_t k = 0; k < limbs.size(); k++) {
        const u64 q = plan->consts[k].q;
        auto key = std::make_pair(q, logn);
        auto it = ctx->tables_a.find(key);
        if (it == ctx->tables_a.end()) {
            // (inserted first and filled in place: tables that were uploaded before a later upload failed stay owned by the cache
            // -- hp_ctx_destroy frees them, the retry fills the rest)
            it = ctx->tables_a.emplace(key, DevTables()).first;
        }
        DevTables &d = it->second;
        if (!d.fwd_ref || !d.inv_ref || !d.fwd_k || !d.inv_k) {
            std::vector<hp::Pair> fwd, inv, fk, ik, t;
            hp::build_fwd_ref(q, logn, fwd);
            hp::build_inv_ref(q, logn, inv);
            hp::build_fwd_fast(fwd, logn, fk);
            hp::build_inv_fast(inv, logn, ik);
            int rc;
            struct { const std::vector<hp::Pair> *src; u64x2 **dst; } parts[4] = {{&fwd, &d.fwd_ref}, {&inv, &d.inv_ref}, {&fk, &d.fwd_k}, {&ik, &d.inv_k}};
            for (auto &pt : parts) {
                if (*pt.dst) continue;
                hp::pairs_to_f64(*pt.src, q, t);
                if ((rc = upload(ctx, t.data(), t.size() * sizeof(hp::Pair), (void **)pt.dst))) return rc;
            }
        }
        HpLimbA &l = limbs[k];
        memset(&l, 0, sizeof(l));
        l.q = (double)q; l.qinv = 1.0 / (double)q; l.qi = q; l.wide = q >= ((u64)1 << 44) ? 1u : 0u;
        l.hi_bound = (u32)((2 * q - 1) >> 32);
        l.range_flag = ctx->sh->range_flag;
        l.fwd_ref = it->second.fwd_ref; l.inv_ref = it->second.inv_ref; l.fwd_k = it->second.fwd_k; l.inv_k = it->second.inv_k;
    }
    HpLimbA *d = nullptr;
    int rc = upload(ctx, limbs.data(), limbs.size() * sizeof(HpLimbA), (void **)&d);
    if (rc) return rc;
    plan->d_limbs_a = d;
    plan->a_state = 1;
    *ok = true;
    return HP_OK;
}
int ensure_plan_a(hp_ctx *ctx, const Plan *plan, bool *ok) {
    return contained(ctx, [&] { return ensure_plan_a_impl(ctx, plan, ok); });
}

int range_check(hp_ctx *ctx) {
    Shared *sh = ctx->sh;
    if (!ctx->a_pending || !sh->range_flag) return HP_OK;
    ctx->a_pending = false;     // (THIS context's stream has just been synchronised; the other members keep their own pending state)
    u32 flag = 0;
    HIP_TRY(ctx, hipMemcpy(&flag, sh->range_flag, sizeof(u32), hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(ctx, hipMemset(sh->range_flag, 0, sizeof(u32)));
        sh->range_trips++;
    }
    if (ctx->trips_seen == sh->range_trips) return HP_OK;
    ctx->trips_seen = sh->range_trips;
    return fail(ctx, HP_ERANGE, "parity level A: an input word was not a lazy word of its limb (>= 2 q): the results of the calls since the "
                                "last synchronisation are not the residues of hehub's words (level B takes any u64)");
}

// A bounded cache of small device objects that is full gets emptied: kernels that may still read the entries are
// drained first.  (Rotation-heavy workloads with thousands of distinct steps would otherwise grow without bound.)
template <class Map> static int make_room(hp_ctx *ctx, Map &m, size_t cap) {
    if (m.size() < cap) return HP_OK;
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (auto &kv : m) (void)hipFree((void *)kv.second);
    m.clear();
    return HP_OK;
}

static int get_cycle_perm_impl(hp_ctx *ctx, size_t logn, size_t step, const u32 **out) {
    const size_t n = (size_t)1 << logn;
    // 3 has order N/2 modulo 2N (N >= 4): steps s and s + N/2 are the same permutation
    const size_t period = logn >= 2 ? n / 2 : 2;
    step %= period;
    auto key = std::make_pair(logn, step);
    auto it = ctx->perms.find(key);
    if (it == ctx->perms.end()) {
        int rc = make_room(ctx, ctx->perms, MAX_PERMS);
        if (rc) return rc;
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
        rc = upload(ctx, perm.data(), n * sizeof(u32), (void **)&d);
        if (rc) return rc;
        it = ctx->perms.emplace(key, d).first;
    }
    *out = it->second;
    return HP_OK;
}
int get_cycle_perm(hp_ctx *ctx, size_t logn, size_t step, const u32 **out) {
    return contained(ctx, [&] { return get_cycle_perm_impl(ctx, logn, step, out); });
}
// room for `count` more maps without the cache being emptied in between: a launch that takes several maps as kernel arguments
// collects their addresses first (hp_dev_ckks_rotate_many)
int reserve_cycle_perms(hp_ctx *ctx, size_t count) {
    return contained(ctx, [&] { return make_room(ctx, ctx->perms, count >= MAX_PERMS ? 1 : MAX_PERMS - count); });
}

// constants of the CRT branch of the many -> one base transform (rns_transform.cpp:86-104), cached per (moduli, t)
int get_crt_consts(hp_ctx *ctx, const uint64_t *moduli, size_t L, u64 t, const HpCrtConsts **out) {
    return contained(ctx, [&] {
        auto key = std::make_pair(std::vector<u64>(moduli, moduli + L), t);
        auto it = ctx->crt.find(key);
        if (it == ctx->crt.end()) {
            int rc = make_room(ctx, ctx->crt, MAX_CRT);
            if (rc) return rc;
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
            rc = upload(ctx, &c, sizeof(c), (void **)&d);
            if (rc) return rc;
            it = ctx->crt.emplace(key, d).first;
        }
        *out = it->second;
        return (int)HP_OK;
    });
}

// The workspace only grows.  Replacing it frees memory that kernels enqueued earlier -- on ANY stream this context was
// pointed at -- may still use, so the whole device is drained first (growth is rare: once per new largest shape).
// A HIP graph captured earlier holds the old pointers: ws_generation tells the caller that it went stale.
int ws_reserve(hp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return HP_OK;
    if (ctx->ws) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    ctx->ws_generation++;
    hipError_t e = hipMalloc(&ctx->ws, bytes);
    if (e != hipSuccess) {
        ctx->ws = nullptr;
        fail(ctx, HP_ENOMEM, std::string("workspace hipMalloc: ") + hipGetErrorString(e));
        return HP_ENOMEM;
    }
    ctx->ws_bytes = bytes;
    return HP_OK;
}

int run_ntt(hp_ctx *ctx, const HpNttJob &job) {
    if (job.W == 0) return HP_OK;
    hipError_t e;
    {
        ProfScope ps(ctx, job.inverse ? "intt" : "ntt");
        if (job.limbs_a) e = hp_launch_ntt_a(job, ctx->stream);
        else if (split_ok(ctx, job.logn, job.W) && !job.pack_mask && !job.pack40_mask) e = hp_launch_ntt_split(job, ctx->stream);   // few limbs: latency
        else if (tiled_ok(ctx, job.logn)) e = hp_launch_ntt_fast(job, ctx->stream);
        else e = hp_launch_ntt_generic(job, ctx->stream);
    }
    if (e != hipSuccess) return fail(ctx, HP_EHIP, std::string("transform launch: ") + hipGetErrorString(e));
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

// Point the context at another stream.  Scratch buffers and cached tables are shared by everything the context
// enqueues, so work enqueued from now on must not overtake what is already queued on the previous stream: one event
// recorded there, waited for here (no host synchronisation).
static int switch_stream(hp_ctx *ctx, hipStream_t to) {
    if (to == ctx->stream) return HP_OK;
    if (!ctx->ev_switch) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_switch, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(to, ctx->ev_switch, 0));
    ctx->stream = to;
    return HP_OK;
}

} // namespace hpi

using namespace hpi;

extern "C" {

const char *hp_version(void) { return "hehub_amd 0.2 (gfx950)"; }

// a context on `device` that belongs to the family `sh` (a fresh one for hp_ctx_create, the parent's for hp_ctx_fork)
static int make_ctx(int device, hpi::Shared *sh, hp_ctx **out) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (hipSetDevice(device) != hipSuccess) return HP_EHIP;
    hp_ctx *c = new (std::nothrow) hp_ctx(sh);
    if (!c) {
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        return HP_ENOMEM;
    }
    c->device = device;
    const hipError_t es = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);   // the caller's current device is not ours to change
    if (es != hipSuccess) {
        delete c;
        return HP_EHIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return HP_OK;
}

int hp_ctx_create(int device, hp_ctx **out) {
    if (!out) return HP_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return HP_EHIP;
    hpi::Shared *sh = new (std::nothrow) hpi::Shared();
    if (!sh) return HP_ENOMEM;
    hp_ctx *c = nullptr;
    const int rc = make_ctx(device, sh, &c);
    if (rc) {
        delete sh;
        return rc;
    }
    c->no_fused_drop = getenv("HP_NO_FUSED_DROP") != nullptr;
    c->no_pack48 = getenv("HP_NO_PACK48") != nullptr;
    c->no_double_drop = getenv("HP_NO_DOUBLE_DROP") != nullptr;
    c->no_pack40 = getenv("HP_NO_PACK40") != nullptr;
    // (every numeric knob is clamped to the range its code path is written for; tests/test_gpu_knobs.py runs the extremes)
    auto clampi = [](long v, long lo, long hi) { return v < lo ? lo : v > hi ? hi : v; };
    if (const char *e = getenv("HP_PACK48_MIN_LOGN")) c->pack48_min_logn = (int)clampi(atol(e), 11, 16);
    c->hks_two_step = getenv("HP_HKS_TWO_STEP") != nullptr;
    c->hks_combine_kernel = getenv("HP_HKS_COMBINE_KERNEL") != nullptr;
    if (const char *e = getenv("HP_DROP_GROUP")) c->drop_group = (int)clampi(atol(e), 0, HP_MAX_LIMBS);
    if (const char *e = getenv("HP_SPREAD_GROUP")) c->spread_group = (int)clampi(atol(e), 0, HP_MAX_LIMBS);   // measured (tools/ab/ab_groups.sh): 4..8 beat 2 by 2-3 % on the launch since the rows are packed; 11 (all moduli) loses it again
    if (const char *e = getenv("HP_SPLIT_MAX_ITEMS")) c->split_max_items = (size_t)clampi(atol(e), 0, 4096);
    if (const char *e = getenv("HP_MULT_STREAMS")) c->mult_streams = atoi(e) >= 2 ? 2 : 1;
    if (const char *e = getenv("HP_MULT_CHUNK")) c->mult_chunk = (size_t)clampi(atol(e), 0, 1l << 30);
    if (const char *e = getenv("HP_PARITY_LEVEL")) c->parity_level = (e[0] == 'A' || e[0] == 'a' || e[0] == '1') ? 1 : 0;
    *out = c;
    return HP_OK;
}

// A second LANE on the parent's GPU: its own stream and scratch workspace, so that its calls overlap on the device with the calls
// of the other members of the family; the twiddle tables, per-chain constants and gather maps are the family's (one copy in HBM and
// in the caches whatever the number of lanes).  Knobs and parity level start as the parent's are now.
int hp_ctx_fork(hp_ctx *parent, hp_ctx **out) {
    if (!parent) return HP_EINVAL;
    hpi::Guard guard__(parent);
    HP_REQUIRE(parent, out);
    hp_ctx *c = nullptr;
    const int rc = make_ctx(parent->device, parent->sh, &c);
    if (rc) return fail(parent, rc, "hp_ctx_fork: could not create the lane's stream");
    parent->sh->refs++;
    c->force_generic = parent->force_generic; c->parity_level = parent->parity_level;
    c->drop_group = parent->drop_group; c->spread_group = parent->spread_group; c->hks_two_step = parent->hks_two_step;
    c->hks_combine_kernel = parent->hks_combine_kernel; c->no_fused_drop = parent->no_fused_drop; c->no_pack48 = parent->no_pack48;
    c->no_pack40 = parent->no_pack40; c->no_double_drop = parent->no_double_drop; c->pack48_min_logn = parent->pack48_min_logn;
    c->mult_streams = parent->mult_streams; c->mult_chunk = parent->mult_chunk; c->split_max_items = parent->split_max_items;
    // the lane will see the calls its parent sees: give it the parent's scratch size now instead of growing to it call by call
    // (a growth drains the device); a failure here is not one of the fork -- the first call reserves what it needs
    // (capped: a parent that has run a batch of 256 holds 10 GiB of scratch, a lane is made for single calls)
    const size_t lane_ws = parent->ws_bytes < ((size_t)256 << 20) ? parent->ws_bytes : ((size_t)256 << 20);
    if (lane_ws) {
        (void)hipSetDevice(parent->device);
        if (hipMalloc(&c->ws, lane_ws) == hipSuccess) c->ws_bytes = lane_ws;
        else { c->ws = nullptr; (void)hipGetLastError(); }
    }
    *out = c;
    return HP_OK;
}

// Device-side ordering between two contexts (normally two lanes of a family, any two contexts of one process work): everything `ctx`
// enqueues from now on runs after everything `other` has enqueued so far.  One event, no host synchronisation.
int hp_ctx_wait_for(hp_ctx *ctx, hp_ctx *other) {
    if (!other) return HP_EINVAL;
    if (ctx == other) return ctx ? HP_OK : HP_EINVAL;
    hipEvent_t ev;
    hipStream_t from;
    {
        HP_ENTER(other);
        if (!other->ev_tail) HIP_TRY(other, hipEventCreateWithFlags(&other->ev_tail, hipEventDisableTiming));
        HIP_TRY(other, hipEventRecord(other->ev_tail, other->stream));
        ev = other->ev_tail;
        from = other->stream;
    }
    HP_ENTER(ctx);
    if (from == ctx->stream) return HP_OK;   // both contexts were pointed at one stream: already ordered
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
    return HP_OK;
}

int hp_ctx_set_parity_level(hp_ctx *ctx, int level) {
    HP_ENTER(ctx);
    if (level != HP_PARITY_B && level != HP_PARITY_A) return fail(ctx, HP_EINVAL, "parity level: HP_PARITY_B (0) or HP_PARITY_A (1)");
    ctx->parity_level = level;
    return HP_OK;
}
int hp_ctx_get_parity_level(hp_ctx *ctx) { return ctx ? ctx->parity_level : HP_EINVAL; }

void hp_ctx_destroy(hp_ctx *ctx) {
    if (!ctx) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    hpi::Shared *sh = ctx->sh;
    bool last;
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        last = --sh->refs == 0;
        if (last) {   // the family's caches go with its last member
            for (auto &kv : sh->tables) {
                (void)hipFree(kv.second.fwd_ref); (void)hipFree(kv.second.inv_ref);
                (void)hipFree(kv.second.fwd_k); (void)hipFree(kv.second.inv_k);
            }
            for (auto &kv : sh->tables_a) {
                (void)hipFree(kv.second.fwd_ref); (void)hipFree(kv.second.inv_ref);
                (void)hipFree(kv.second.fwd_k); (void)hipFree(kv.second.inv_k);
            }
            for (auto &kv : sh->plans) { (void)hipFree(kv.second.d_limbs); if (kv.second.d_limbs_a) (void)hipFree(kv.second.d_limbs_a); }
            for (auto &kv : sh->perms) (void)hipFree(kv.second);
            for (auto &kv : sh->crt) (void)hipFree(kv.second);
            for (auto &kv : sh->hks) (void)hipFree(kv.second);
            if (sh->range_flag) (void)hipFree(sh->range_flag);
        }
        if (ctx->ws) (void)hipFree(ctx->ws);
        for (auto &ev : ctx->prof_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
        for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
        for (int i = 0; i < 2; i++) {
            if (ctx->aux[i]) (void)hipStreamDestroy(ctx->aux[i]);
            if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
        }
        if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
        if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
        if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
        (void)hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
    if (last) delete sh;
}

// The calling thread's last failure on this context if it had one, else the context's last failure.  The pointer
// stays valid until the calling thread's next hp_* call on any context.
const char *hp_last_error(hp_ctx *ctx) {
    if (!ctx) return "null context";
    if (tl_err_ctx == ctx) return tl_err_msg.c_str();
    std::lock_guard<std::mutex> lk(ctx->mu);
    tl_err_msg = ctx->err;
    tl_err_ctx = ctx;
    return tl_err_msg.c_str();
}

int hp_ctx_set_stream(hp_ctx *ctx, void *s) {
    HP_ENTER(ctx);
    return switch_stream(ctx, (hipStream_t)s);   // NULL is the HIP default (null) stream, e.g. torch's default stream
}
int hp_ctx_reset_stream(hp_ctx *ctx) {
    HP_ENTER(ctx);
    return switch_stream(ctx, ctx->own_stream);
}
void *hp_ctx_get_stream(hp_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// The scratch workspace only grows (the largest call so far defines it: 10+ GiB for a C3 batch of 256); this gives it back.
int hp_ctx_release_workspace(hp_ctx *ctx) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipDeviceSynchronize());   // every stream that may have used it, not only the current one
    if (ctx->ws) HIP_TRY(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->ws_generation++;
    return HP_OK;
}
size_t hp_ctx_workspace_bytes(hp_ctx *ctx) { return ctx ? ctx->ws_bytes : 0; }
unsigned long hp_ctx_workspace_generation(hp_ctx *ctx) { return ctx ? ctx->ws_generation : 0; }

int hp_sync(hp_ctx *ctx) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return range_check(ctx);
}

int hp_dev_alloc(hp_ctx *ctx, size_t bytes, void **dptr) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, dptr);
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return fail(ctx, HP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return HP_OK;
}
int hp_dev_free(hp_ctx *ctx, void *dptr) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(dptr));
    return HP_OK;
}
int hp_host_alloc(hp_ctx *ctx, size_t bytes, void **hptr) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, hptr);
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) return fail(ctx, HP_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return HP_OK;
}
int hp_host_free(hp_ctx *ctx, void *hptr) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostFree(hptr));
    return HP_OK;
}
int hp_memcpy_h2d(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HP_OK;
}
int hp_memcpy_d2h(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return range_check(ctx);   // (the words are there either way; the status says whether a level-A call that made them was in range)
}
// Host memory the CALLER owns, made DMA-able in place (hipHostRegister): transfers from / to it then run at the link rate and
// asynchronously, without the runtime's internal staging of pageable memory.
int hp_host_register(hp_ctx *ctx, void *hptr, size_t bytes) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, hptr);
    hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // a range that lies inside pages an earlier registration already covers (two heap blocks on one page, a block that was
        // registered as part of a larger one) is as DMA-able as it gets: that is a success, anything else is not
        // (EVERY page of the range is asked for: two registrations at its ends with unregistered pages between them are not a cover)
        if (e == hipErrorHostMemoryAlreadyRegistered && bytes > 0) {
            bool covered = true;
            const uintptr_t first = (uintptr_t)hptr, last = first + bytes - 1;
            void *d = nullptr;
            for (uintptr_t a = first; covered && a <= last; a = (a | (uintptr_t)4095) + 1)
                covered = hipHostGetDevicePointer(&d, (void *)a, 0) == hipSuccess;
            covered = covered && hipHostGetDevicePointer(&d, (void *)last, 0) == hipSuccess;
            if (covered) return HP_OK;
        }
        (void)hipGetLastError();
        return fail(ctx, HP_EHIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    }
    return HP_OK;
}
int hp_host_unregister(hp_ctx *ctx, void *hptr) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, hptr);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostUnregister(hptr));
    return HP_OK;
}
// enqueue only: the host buffer must stay untouched (h2d: unmodified; d2h: unread) until hp_sync() or a synchronous call returns
int hp_memcpy_h2d_async(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return HP_OK;
}
int hp_ctx_device(hp_ctx *ctx) { return ctx ? ctx->device : HP_EINVAL; }
int hp_memcpy_peer_async(hp_ctx *dst_ctx, void *dst, hp_ctx *src_ctx, const void *src, size_t bytes) {
    if (!src_ctx) return HP_EINVAL;
    const int src_dev = src_ctx->device;   // (set once when the context is made)
    HP_ENTER(dst_ctx);
    HP_REQUIRE(dst_ctx, dst, src);
    if (bytes == 0) return HP_OK;
    if (src_dev == dst_ctx->device) {
        HIP_TRY(dst_ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dst_ctx->stream));
        return HP_OK;
    }
    // direct access for the pair, once (both directions are asked for by whoever copies that way first); a refusal is not an error:
    // hipMemcpyPeerAsync then goes through the runtime's staging
    static std::mutex pair_mu;
    static std::map<std::pair<int, int>, bool> asked;
    {
        std::lock_guard<std::mutex> lk(pair_mu);
        bool &done = asked[{dst_ctx->device, src_dev}];
        if (!done) {
            done = true;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dst_ctx->device, src_dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src_dev, 0);
            (void)hipGetLastError();
        }
    }
    HIP_TRY(dst_ctx, hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_dev, bytes, dst_ctx->stream));
    return HP_OK;
}
int hp_memcpy_d2h_async(hp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    HP_ENTER(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return HP_OK;
}
// One polynomial between contiguous device rows and separate registered host blocks, by ONE kernel over PCIe (enqueue only).
static int host_rows(hp_ctx *ctx, bool to_host, size_t rows, size_t words, u64 *dev, void *const *h_rows) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, dev, h_rows);
    HP_ALIGNED(ctx, dev);
    if (words & 1) return fail(ctx, HP_EINVAL, "host rows: an even number of words per row");
    for (size_t r0 = 0; r0 < rows; r0 += HP_HOST_ROWS_MAX) {
        const size_t cnt = rows - r0 < HP_HOST_ROWS_MAX ? rows - r0 : HP_HOST_ROWS_MAX;
        HpHostRows hr;
        for (size_t r = 0; r < cnt; r++) {
            if (!h_rows[r0 + r] || ((uintptr_t)h_rows[r0 + r] & 15u)) return fail(ctx, HP_EINVAL, "host rows: NULL or misaligned row");
            void *d = nullptr;
            hipError_t e = hipHostGetDevicePointer(&d, h_rows[r0 + r], 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, HP_EINVAL, "host rows: a row is not registered host memory (hp_host_register / hp_host_alloc)");
            }
            hr.p[r] = (u64 *)d;
        }
        ProfScope ps(ctx, "copy");
        int rc = chk(ctx, hp_launch_host_rows(to_host, hr, (u32)cnt, words, dev + r0 * words, ctx->stream), "host rows");
        if (rc) return rc;
    }
    return HP_OK;
}
int hp_dev_store_host_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *d_src, uint64_t *const *h_rows) {
    return host_rows(ctx, true, rows, words, const_cast<u64 *>(d_src), (void *const *)h_rows);
}
int hp_dev_load_host_rows(hp_ctx *ctx, size_t rows, size_t words, uint64_t *d_dst, const uint64_t *const *h_rows) {
    return host_rows(ctx, false, rows, words, d_dst, (void *const *)h_rows);
}
// Rows that lie anywhere in device memory <-> one packed block u64[rows][words]: the batch entry points want [batch][2][L][N], the
// ciphertexts of an application are separate objects.  One kernel per 64 rows (the row pointers travel as kernel arguments).
static int dev_rows(hp_ctx *ctx, bool scatter, size_t rows, size_t words, u64 *packed, void *const *d_rows) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, packed, d_rows);
    HP_ALIGNED(ctx, packed);
    if (words & 1) return fail(ctx, HP_EINVAL, "device rows: an even number of words per row");
    for (size_t r0 = 0; r0 < rows; r0 += HP_HOST_ROWS_MAX) {
        const size_t cnt = rows - r0 < HP_HOST_ROWS_MAX ? rows - r0 : HP_HOST_ROWS_MAX;
        HpHostRows hr;
        for (size_t r = 0; r < cnt; r++) {
            if (!d_rows[r0 + r] || ((uintptr_t)d_rows[r0 + r] & 15u)) return fail(ctx, HP_EINVAL, "device rows: NULL or misaligned row");
            hr.p[r] = (u64 *)d_rows[r0 + r];
        }
        ProfScope ps(ctx, "copy");
        int rc = chk(ctx, hp_launch_host_rows(scatter, hr, (u32)cnt, words, packed + r0 * words, ctx->stream), "device rows");
        if (rc) return rc;
    }
    return HP_OK;
}
int hp_dev_gather_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *const *d_rows, uint64_t *d_dst) {
    return dev_rows(ctx, false, rows, words, d_dst, (void *const *)d_rows);
}
int hp_dev_scatter_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *d_src, uint64_t *const *d_rows) {
    return dev_rows(ctx, true, rows, words, const_cast<u64 *>(d_src), (void *const *)d_rows);
}
int hp_ctx_set_force_generic(hp_ctx *ctx, int on) {
    HP_ENTER(ctx);
    ctx->force_generic = on != 0;
    return HP_OK;
}

} // extern "C"
