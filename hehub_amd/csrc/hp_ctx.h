// hp_ctx.h -- engine-internal declarations shared by the C-ABI translation units:
//   hp_ctx.cpp         context life cycle, streams, workspace, table / plan / gather-map caches, error strings
//   hp_prof.cpp        in-library kernel timing (HIP events on the launch stream)
//   hp_api_poly.cpp    drop-in host entry points, device-resident polynomial batches, encrypt / decrypt cores, base conversions
//   hp_api_scheme.cpp  key switch, drop-last-prime, relinearisation, rotations, the mult pipelines, limb-range stages
//   hp_api_hks.cpp     hybrid key switch (extension)
//   hp_node.cpp        several contexts (GPUs) behind one handle: batch slices and the limb-sharded exchange
// There is NO CPU fallback anywhere: every entry point launches HIP kernels or fails.
#pragma once
#include "../../include/hehub_amd.h"

#include "hp_kernels.h"
#include "hp_tables.h"

#include <hip/hip_runtime.h>

#include <initializer_list>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace hpi {

struct DevTables {
    u64x2 *fwd_ref = nullptr, *inv_ref = nullptr, *fwd_k = nullptr, *inv_k = nullptr;
};

struct Plan {
    HpLimb *d_limbs = nullptr;
    std::vector<hp::ModConsts> consts;
    // parity level A: the FP64 per-limb records, built on first use (ensure_plan_a); a_state 0 = not tried, 1 = ready,
    // -1 = not available for this chain (a modulus >= 2^50, or a ring degree without tiled kernels)
    mutable HpLimbA *d_limbs_a = nullptr;
    mutable int a_state = 0;
    size_t logn = 0;
};

struct ProfEvent {
    hipEvent_t a, b;
    const char *family;   // a string literal of the launch site
};

// caches of small device objects are bounded: when one is full it is emptied (after a device synchronise), not grown
constexpr size_t MAX_PERMS = 1024, MAX_CRT = 128, MAX_HKS = 16;

} // namespace hpi

namespace hpi {
// What the contexts of one FAMILY have in common (hp_ctx_create makes a family of one, hp_ctx_fork adds a lane to it): the device,
// ONE lock -- every entry point of every member runs under it, so the host side of a family is serialised exactly as a single
// context's always was, while the members' streams overlap on the device -- and the caches of immutable device objects (twiddle
// tables, per-chain constants, gather maps), which all members read.
struct Shared {
    std::mutex mu;
    int refs = 1;
    std::map<std::pair<u64, size_t>, DevTables> tables;          // (q, logn)
    std::map<std::pair<u64, size_t>, DevTables> tables_a;        // (q, logn): the same tables as (w, w / q) doubles (parity level A)
    std::map<std::pair<size_t, std::vector<u64>>, Plan> plans;   // (logn, moduli); logn == 0: no transforms needed
    std::map<std::pair<size_t, size_t>, u32 *> perms;                 // (logn, step mod N/2) -> gather map
    std::map<std::pair<std::vector<u64>, u64>, HpCrtConsts *> crt;    // (old moduli, new modulus) -> CRT-branch constants
    std::map<std::pair<std::vector<u64>, std::pair<size_t, size_t>>, HpHksConsts *> hks;   // (extended moduli, (k, alpha))
    // level A range guard (hp_ntt_a.hip: RangeAcc): one sticky device word for the family.  Every member keeps its OWN "a level-A call
    // of mine has not been checked yet" state (hp_ctx::a_pending) and reads the word after a synchronisation of its OWN stream; a
    // non-zero word is counted here (`range_trips`) and cleared, and every member that had level-A work pending when a trip was
    // counted reports HP_ERANGE once -- whichever lane synchronises first (range_check, hp_ctx.cpp).
    u32 *range_flag = nullptr;
    unsigned long range_trips = 0;
};
} // namespace hpi

struct hp_ctx {
    hpi::Shared *sh;
    explicit hp_ctx(hpi::Shared *s)
        : sh(s), mu(s->mu), tables(s->tables), tables_a(s->tables_a), plans(s->plans), perms(s->perms), crt(s->crt), hks(s->hks) {}
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_switch = nullptr;   // orders a newly selected stream after the work already enqueued on the previous one
    hipEvent_t ev_tail = nullptr;     // hp_ctx_wait_for: marks "everything this context has enqueued so far" for another one to wait on
    std::mutex &mu;                   // the family's lock
    std::string err;
    bool force_generic = false;
    // parity level of the scheme-level pipelines (hp_ctx_set_parity_level; HP_PARITY_LEVEL=A): 0 = B, raw words identical to
    // hehub's (default); 1 = A, canonical residues through the FP64 transforms of hp_ntt_a.hip where the chain allows
    int parity_level = 0;
    bool cur_a = false;           // set for the duration of one entry point (under the context lock): this call runs at level A
    bool a_pending = false;       // a level-A call has been enqueued on THIS context since its last range_check
    unsigned long trips_seen = 0; // the family's range_trips when a_pending was set / last reported
    // the family's caches (hpi::Shared)
    std::map<std::pair<u64, size_t>, hpi::DevTables> &tables;
    std::map<std::pair<u64, size_t>, hpi::DevTables> &tables_a;
    std::map<std::pair<size_t, std::vector<u64>>, hpi::Plan> &plans;
    std::map<std::pair<size_t, size_t>, u32 *> &perms;
    std::map<std::pair<std::vector<u64>, u64>, HpCrtConsts *> &crt;
    std::map<std::pair<std::vector<u64>, std::pair<size_t, size_t>>, HpHksConsts *> &hks;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    unsigned long ws_generation = 0;   // bumped whenever the workspace is reallocated or released (captured graphs go stale)
    // profiling
    std::string prof_family;
    bool prof_on = false;
    std::vector<hpi::ProfEvent> prof_events;
    std::vector<hipEvent_t> event_pool;
    // two auxiliary streams for software-pipelined sub-batches (dev_mult)
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_start = nullptr, ev_done[2] = {nullptr, nullptr};
    // knobs, read from the environment once when the context is created (all default to the measured-best setting)
    int drop_group = 4;           // HP_DROP_GROUP=G: the same numbering for the fused drop launch (every limb reads one coefficient row)
    int spread_group = 4;         // HP_SPREAD_GROUP=G: digit-spread launch numbered by groups of G moduli (0: modulus-major)
    bool hks_two_step = false;    // HP_HKS_TWO_STEP: hybrid mult = switch, then a separate rescale (instead of the merged transform)
    bool hks_combine_kernel = false;   // HP_HKS_COMBINE_KERNEL: the merged ModDown + rescale combination as its own kernel
    bool no_fused_drop = false;   // HP_NO_FUSED_DROP: separate drop_rem / NTT / drop_fin launches
    bool no_pack48 = false;       // HP_NO_PACK48: digit workspace always as plain u64 rows
    bool no_pack40 = false;       // HP_NO_PACK40: level A keeps 48-bit digit rows for the moduli below 2^40
    bool no_double_drop = false;  // HP_NO_DOUBLE_DROP: level A keeps relinearize's mod-down and the rescale / mod switch of a mult as two launches
    int pack48_min_logn = 11;     // HP_PACK48_MIN_LOGN: smallest ring degree whose digit rows are packed
    size_t split_max_items = 128; // HP_SPLIT_MAX_ITEMS: a level-B transform launch of at most this many limbs (N >= 4096) runs split over
                                  // N / 2048 workgroups per limb (hp_ntt_split.hip: a quarter of the latency, a third of the throughput); 0 = never
    int mult_streams = 1;         // HP_MULT_STREAMS=2: software-pipeline two sub-batches in dev_mult
    size_t mult_chunk = 0;        // HP_MULT_CHUNK: sub-batch size of dev_mult (0 = whole batch, or half with 2 streams)
};

namespace hpi {

// ---- errors -----------------------------------------------------------------------------------------------
// The message goes to the context AND to a per-thread slot, so that hp_last_error() on the failing thread returns the
// message of ITS call even when another thread fails on the same context in between.
int fail(hp_ctx *ctx, int code, const std::string &msg);

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) return hpi::fail((ctx), HP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

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
    if (hpi::any_null({__VA_ARGS__})) return hpi::fail(ctx, HP_EINVAL, "NULL pointer argument")
// the kernels move 16 bytes per lane: device operands must be 16-byte aligned (every allocator's blocks are)
inline bool any_misaligned(std::initializer_list<const void *> ptrs) {
    for (const void *p : ptrs)
        if ((uintptr_t)p & 15u) return true;
    return false;
}
#define HP_ALIGNED(ctx, ...) \
    if (hpi::any_misaligned({__VA_ARGS__})) return hpi::fail(ctx, HP_EINVAL, "device pointers must be 16-byte aligned")

// every entry point runs with the context's device current and puts the caller's device back when it returns: an application
// that drives several GPUs from one thread (or uses torch next to the engine) keeps allocating where it thinks it does
struct Guard {
    hp_ctx *ctx;
    std::unique_lock<std::mutex> lk;
    int prev = -1;
    explicit Guard(hp_ctx *c) : ctx(c), lk(c->mu) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) (void)hipSetDevice(c->device);
    }
    ~Guard() {
        if (prev >= 0 && prev != ctx->device) (void)hipSetDevice(prev);
    }
};
// first statement of every entry point: a NULL context is an argument error, not a crash; then the context lock
#define HP_ENTER(ctx)                 \
    if (!(ctx)) return HP_EINVAL;     \
    hpi::Guard guard__(ctx)

int chk(hp_ctx *ctx, hipError_t e, const char *what);
int fail_local(hp_ctx *ctx, int code, const std::string &msg);   // without the context lock: calling thread's slot only
int chk_local(hp_ctx *ctx, hipError_t e, const char *what);
int upload(hp_ctx *ctx, const void *host, size_t bytes, void **dptr);

// ---- caches -----------------------------------------------------------------------------------------------
int get_tables(hp_ctx *ctx, u64 q, size_t logn, DevTables &out);
// device array of per-limb constants for a modulus chain; with_ntt == false skips the twiddles
int get_plan(hp_ctx *ctx, size_t logn, const uint64_t *moduli, size_t count, bool with_ntt, const Plan **out);
// level-A records of a plan (built once); *ok = false when the chain / ring degree has no level-A kernels (the call then runs at B)
int ensure_plan_a(hp_ctx *ctx, const Plan *plan, bool *ok);
// after a host synchronisation of ctx's stream: HP_ERANGE when a level-A kernel of the family saw a word outside its range while
// this context had level-A work that had not been checked (per context: a lane that synchronises first does not use up the report
// of a lane whose kernels are still running)
int range_check(hp_ctx *ctx);
// a level-A call is being enqueued on ctx: its next synchronising entry point looks at the family's range word
inline void mark_level_a(hp_ctx *c) {
    if (!c->a_pending) {
        c->a_pending = true;
        c->trips_seen = c->sh->range_trips;   // (trips counted before this context had anything at stake are not its concern)
    }
}
// first thing a scheme-level entry point does after get_plan: decides whether THIS call runs at level A
struct LevelScope {
    hp_ctx *ctx;
    int rc = 0;
    LevelScope(hp_ctx *c, const Plan *plan) : ctx(c) {
        c->cur_a = false;
        // (level A is the tiled FP64 kernels end to end: not with the debug switches that route a call through the simple
        // kernels or the unfused drop -- a mix of representatives would satisfy neither level's contract)
        if (c->parity_level == 1 && plan && !c->force_generic && !c->no_fused_drop) {
            bool ok = false;
            rc = ensure_plan_a(c, plan, &ok);
            c->cur_a = (rc == 0) && ok;
            if (c->cur_a) mark_level_a(c);
        }
    }
    ~LevelScope() { ctx->cur_a = false; }
};
// gather map of cycle(poly, step) (permutation.cpp:39-53): out[to] = in[perm[to]]
int get_cycle_perm(hp_ctx *ctx, size_t logn, size_t step, const u32 **out);
int reserve_cycle_perms(hp_ctx *ctx, size_t count);   // the next `count` get_cycle_perm calls do not empty the cache
int get_crt_consts(hp_ctx *ctx, const uint64_t *moduli, size_t L, u64 t, const HpCrtConsts **out);

// ---- workspace --------------------------------------------------------------------------------------------
// grow-only scratch; stream order makes reuse across calls safe (hp_ctx_set_stream orders streams, see hp_ctx.cpp)
int ws_reserve(hp_ctx *ctx, size_t bytes);
inline size_t padded(size_t words) { return (words * 8 + 255) & ~(size_t)255; }
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    u64 *take(size_t words) {
        u64 *p = (u64 *)(base + off);
        off += padded(words);
        return p;
    }
};

// ---- profiling brackets (hp_prof.cpp) -----------------------------------------------------------------------
struct ProfScope {
    hp_ctx *ctx;
    bool on;
    ProfEvent ev;
    ProfScope(hp_ctx *c, const char *family);
    ~ProfScope();
};

// ---- transforms -------------------------------------------------------------------------------------------
inline bool logn_ok(size_t logn) { return logn >= 1 && logn <= 16; }   // (ntt.cpp:26-29 takes any degree with 2N | q - 1; the reference's bit reversal is 16 bits wide, permutation.h:41-55)
inline bool tiled_ok(const hp_ctx *ctx, size_t logn) { return !ctx->force_generic && logn >= 11 && logn <= 15; }
inline bool fused_drop_ok(const hp_ctx *ctx, size_t logn) { return tiled_ok(ctx, logn) && !ctx->no_fused_drop; }
// a launch of `items` limb transforms small enough for the split (latency) path
inline bool split_ok(const hp_ctx *ctx, size_t logn, size_t items) {
    return !ctx->force_generic && logn >= 12 && logn <= 16 && items > 0 && items <= ctx->split_max_items;
}
#define HP_LOGN_MSG "ring degrees 2^1 .. 2^16 are supported"
int run_ntt(hp_ctx *ctx, const HpNttJob &job);
HpNttJob batch_job(const Plan *plan, size_t logn, size_t L, size_t P, const u64 *src, u64 *dst, size_t src_ps, size_t dst_ps,
                   int inverse, int strict);

// ---- building blocks of the scheme-level pipelines (hp_api_scheme.cpp), also used by the hybrid key switch ----
size_t ext_prod_ws_words(size_t n, size_t L, size_t P);
size_t drop_ws_words(size_t n, size_t L, size_t P2);
int ks_coef(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P, size_t j0, size_t j1, const u64 *pt, size_t pt_pstride,
            u64 *coef, bool for_spread_a = false);
int drop_last(hp_ctx *ctx, const Plan *plan, size_t logn, size_t L, size_t P2, bool bgv, u64 t, const u64 *x, const u64 *addend,
              size_t add_poly_stride, size_t add_ct_stride, u32 add_mask, u64 *out, Carver &cv);

} // namespace hpi
