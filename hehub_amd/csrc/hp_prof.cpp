// hp_prof.cpp -- in-library kernel timing: between hp_prof_begin and hp_prof_end every launch of the named kernel
// family is bracketed by hipEvents recorded on the stream the kernel is launched on (bench.py's roofline numbers).
#include "hp_ctx.h"

namespace hpi {

static hipEvent_t get_event(hp_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;   // the scope then stays off (ProfScope checks)
    return e;
}

ProfScope::ProfScope(hp_ctx *c, const char *family) : ctx(c) {
    on = c->prof_on && (c->prof_family == family || c->prof_family == "*");   // "*": every family (hp_prof_end_families)
    ev.family = family;
    if (on) {
        ev.a = get_event(c);
        ev.b = get_event(c);
        if (!ev.a || !ev.b) {   // no events to be had: this launch goes untimed
            if (ev.a) c->event_pool.push_back(ev.a);
            if (ev.b) c->event_pool.push_back(ev.b);
            on = false;
            return;
        }
        (void)hipEventRecord(ev.a, c->stream);
    }
}
ProfScope::~ProfScope() {
    if (on) {
        (void)hipEventRecord(ev.b, ctx->stream);
        ctx->prof_events.push_back(ev);
    }
}

} // namespace hpi

using namespace hpi;

extern "C" {

int hp_prof_begin(hp_ctx *ctx, const char *family) {
    HP_ENTER(ctx);
    return contained(ctx, [&] {
        for (auto &ev : ctx->prof_events) { ctx->event_pool.push_back(ev.a); ctx->event_pool.push_back(ev.b); }
        ctx->prof_events.clear();
        ctx->prof_family = family ? family : "";
        ctx->prof_on = true;
        return (int)HP_OK;
    });
}

// launches and milliseconds per family of the bracketed launches, in order of first appearance
static int prof_collect(hp_ctx *ctx, std::vector<std::pair<const char *, std::pair<size_t, double>>> &fam) {
    ctx->prof_on = false;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // every event goes back to the pool exactly once and the list is emptied BEFORE any error return (an event left in both
    // would be handed to two scopes and destroyed twice)
    hipError_t bad = hipSuccess;
    for (auto &ev : ctx->prof_events) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, ev.a, ev.b);
        if (e != hipSuccess && bad == hipSuccess) bad = e;
        size_t i = 0;
        while (i < fam.size() && std::string(fam[i].first) != ev.family) i++;
        if (i == fam.size()) fam.push_back({ev.family, {0, 0.0}});
        fam[i].second.first++;
        fam[i].second.second += ms;
        ctx->event_pool.push_back(ev.a);
        ctx->event_pool.push_back(ev.b);
    }
    ctx->prof_events.clear();
    if (bad != hipSuccess) return fail(ctx, HP_EHIP, std::string("hipEventElapsedTime: ") + hipGetErrorString(bad));
    return HP_OK;
}

int hp_prof_end(hp_ctx *ctx, size_t *launches, double *total_ms) {
    HP_ENTER(ctx);
    return contained(ctx, [&] {
        std::vector<std::pair<const char *, std::pair<size_t, double>>> fam;
        const int rc = prof_collect(ctx, fam);
        if (rc) return rc;
        size_t count = 0;
        double total = 0;
        for (auto &f : fam) { count += f.second.first; total += f.second.second; }
        if (launches) *launches = count;
        if (total_ms) *total_ms = total;
        return (int)HP_OK;
    });
}

int hp_prof_end_families(hp_ctx *ctx, size_t cap, const char **names, size_t *launches, double *total_ms, size_t *count) {
    HP_ENTER(ctx);
    HP_REQUIRE(ctx, names, launches, total_ms, count);
    return contained(ctx, [&] {
        std::vector<std::pair<const char *, std::pair<size_t, double>>> fam;
        const int rc = prof_collect(ctx, fam);
        if (rc) return rc;
        *count = fam.size() < cap ? fam.size() : cap;
        for (size_t i = 0; i < *count; i++) {
            names[i] = fam[i].first;
            launches[i] = fam[i].second.first;
            total_ms[i] = fam[i].second.second;
        }
        return (int)HP_OK;
    });
}

} // extern "C"
