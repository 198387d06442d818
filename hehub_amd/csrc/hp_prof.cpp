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
    on = c->prof_on && c->prof_family == family;
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

int hp_prof_end(hp_ctx *ctx, size_t *launches, double *total_ms) {
    HP_ENTER(ctx);
    ctx->prof_on = false;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // every event goes back to the pool exactly once and the list is emptied BEFORE any error return (an event left in both
    // would be handed to two scopes and destroyed twice)
    double total = 0;
    hipError_t bad = hipSuccess;
    const size_t count = ctx->prof_events.size();
    for (auto &ev : ctx->prof_events) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, ev.a, ev.b);
        if (e != hipSuccess && bad == hipSuccess) bad = e;
        total += ms;
        ctx->event_pool.push_back(ev.a);
        ctx->event_pool.push_back(ev.b);
    }
    ctx->prof_events.clear();
    if (bad != hipSuccess) return fail(ctx, HP_EHIP, std::string("hipEventElapsedTime: ") + hipGetErrorString(bad));
    if (launches) *launches = count;
    if (total_ms) *total_ms = total;
    return HP_OK;
}

} // extern "C"
