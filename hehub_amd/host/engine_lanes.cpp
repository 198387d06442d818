// engine_lanes.cpp -- the process-wide engine of the host layer: device ranks and lanes, which call runs on which slot, the ordering of
// dependent calls across slots (one event each), and the mapping of engine status codes onto hehub's exceptions.  (layer.hpp has the map
// of the layer's files.)
#include "layer.hpp"

namespace hehub {

namespace amd {

LaneSet &lane_set() {
    static LaneSet &S = *new LaneSet;   // (never destroyed: see Pool)
    return S;
}

hp_ctx *engine() {
    static std::once_flag once;
    LaneSet &S = lane_set();
    std::call_once(once, [&S] {
        int dev = 0;
        if (const char *e = std::getenv("HEHUB_AMD_DEVICE")) dev = std::atoi(e);
        S.devs[0] = dev;
#ifndef HEHUB_AMD_BIND_REFERENCE
        // HEHUB_AMD_DEVICES=<n>: HIP devices 0 .. n-1; HEHUB_AMD_DEVICES=<a>,<b>,...: those devices, one rank each (a device may
        // appear more than once: ranks then share it)
        if (const char *e = std::getenv("HEHUB_AMD_DEVICES")) {
            std::vector<int> list;
            if (std::strchr(e, ',')) {
                for (const char *p = e; *p;) {
                    char *end = nullptr;
                    const long d = std::strtol(p, &end, 10);
                    if (end == p) break;
                    list.push_back((int)d);
                    p = *end == ',' ? end + 1 : end;
                }
            } else {
                for (int d = 0; d < std::atoi(e); d++) list.push_back(d);
            }
            if (!list.empty()) {
                S.ndev = (int)std::min(list.size(), (size_t)MAX_DEVS);
                for (int r = 0; r < S.ndev; r++) S.devs[r] = list[r];
            }
        }
#endif
        if (hp_ctx_create(S.devs[0], &S.v[0].ctx) != HP_OK) S.v[0].ctx = nullptr;
#ifndef HEHUB_AMD_BIND_REFERENCE
        S.count = 4;
        if (const char *e = std::getenv("HEHUB_AMD_LANES")) S.count = std::max(1, std::min(MAX_LANES, std::atoi(e)));
#endif
        if (S.v[0].ctx) S.level_a = hp_ctx_get_parity_level(S.v[0].ctx) == HP_PARITY_A;
    });
    if (!S.v[0].ctx) throw std::runtime_error("hehub_amd: no MI355X engine available (hp_ctx_create failed); there is no CPU fallback");
    return S.v[0].ctx;
}

// the context of the lane the current call runs on
hp_ctx *cur() {
    (void)engine();
    LaneSet &S = lane_set();
    Lane &L = S.v[S.cur];
    if (!L.ctx) {
        const int rank = rank_of(S.cur);
        Lane &R = S.v[rank * MAX_LANES];   // the rank's root context: its own engine family (tables, plans) on its own device
        if (!R.ctx) {
            if (hp_ctx_create(S.devs[rank], &R.ctx) != HP_OK) {
                R.ctx = nullptr;
                throw std::runtime_error("hehub_amd: no engine on HIP device " + std::to_string(S.devs[rank]) + " (rank " + std::to_string(rank) + " of HEHUB_AMD_DEVICES)");
            }
            (void)hp_ctx_set_parity_level(R.ctx, S.level_a ? HP_PARITY_A : HP_PARITY_B);
        }
        if (!L.ctx) {   // (a lane of the rank: a fork of its root -- own stream and scratch, the family's tables)
            if (hp_ctx_fork(R.ctx, &L.ctx) != HP_OK) throw std::runtime_error(std::string("hehub_amd: ") + hp_last_error(R.ctx));
            (void)hp_ctx_set_parity_level(L.ctx, S.level_a ? HP_PARITY_A : HP_PARITY_B);
        }
    }
    return L.ctx;
}
int cur_rank() { return rank_of(lane_set().cur); }
// the root context of a device rank (made on first use)
hp_ctx *rank_ctx(int rank) {
    LaneSet &S = lane_set();
    if (!S.v[rank * MAX_LANES].ctx) {
        const int saved = S.cur;
        S.cur = rank * MAX_LANES;
        try {
            (void)cur();
        } catch (...) {
            S.cur = saved;
            throw;
        }
        S.cur = saved;
    }
    return S.v[rank * MAX_LANES].ctx;
}

int lanes() {
    (void)engine();
    return lane_set().count;
}
void synchronize() {
    (void)engine();
    flush_all();
    LaneSet &S = lane_set();
    for (int l = 0; l < MAX_SLOTS; l++) {
        if (S.v[l].ctx && hp_sync(S.v[l].ctx) != HP_OK) throw std::runtime_error(std::string("hehub_amd: ") + hp_last_error(S.v[l].ctx));
        S.v[l].busy = false;
    }
}
// everything every slot has enqueued is done: every slot has "seen" every other one up to now, so the read / write records that blocks
// still carry from before (also of lanes / ranks that go out of use) never make anybody wait again
[[maybe_unused]] static void all_seen() {
    LaneSet &S = lane_set();
    for (int a = 0; a < MAX_SLOTS; a++)
        for (int b = 0; b < MAX_SLOTS; b++) S.v[a].seen[b] = S.v[b].ticket;
}
void set_lanes(int n) {
#ifdef HEHUB_AMD_BIND_REFERENCE
    (void)n;
#else
    synchronize();   // (a lane that goes out of use must not owe anybody anything)
    LaneSet &S = lane_set();
    all_seen();
    S.count = std::max(1, std::min(MAX_LANES, n));
    S.cur = 0;
    for (int r = 0; r < MAX_DEVS; r++) S.rr[r] = 0;
    S.last = -1;
#endif
}
int devices() {
    (void)engine();
    return lane_set().ndev;
}
void set_devices(const std::vector<int> &hip_devices) {
#ifdef HEHUB_AMD_BIND_REFERENCE
    (void)hip_devices;   // (hehub's own objects are host memory: one device, every call ends with a download)
#else
    if (hip_devices.empty() || hip_devices.size() > (size_t)MAX_DEVS) throw std::invalid_argument("hehub_amd: between 1 and 8 device ranks");
    synchronize();
    LaneSet &S = lane_set();
    // a rank that has been used keeps its device: blocks, keys and tables made there stay valid (a rank that goes out of use keeps its
    // contexts; what lives there moves over when a call needs it)
    for (size_t r = 0; r < hip_devices.size(); r++)
        if (S.v[r * MAX_LANES].ctx && S.devs[r] != hip_devices[r])
            throw std::logic_error("hehub_amd: device rank " + std::to_string(r) + " is already in use on another HIP device");
    all_seen();
    S.ndev = (int)hip_devices.size();
    for (int r = 0; r < S.ndev; r++) S.devs[r] = hip_devices[r];
    S.cur = 0;
    S.rr_dev = -1;
    S.last = -1;
#endif
}
void set_devices(int n) {
    std::vector<int> list;
    for (int d = 0; d < n; d++) list.push_back(d);
    set_devices(list);
}

// parity level of the process-wide engine (include/hehub_amd.h: hp_ctx_set_parity_level): false = B, hehub's raw lazy words (default);
// true = A, the scheme-level calls return canonical residues (reduce_strict of hehub's words) through the FP64 transforms
void set_parity_level_a(bool on) {
    (void)engine();
    flush_all();   // (recorded calls run at the level that was set when they were recorded)
    LaneSet &S = lane_set();
    for (int l = 0; l < MAX_SLOTS; l++)
        if (S.v[l].ctx && hp_ctx_set_parity_level(S.v[l].ctx, on ? HP_PARITY_A : HP_PARITY_B) != HP_OK)
            throw std::runtime_error(hp_last_error(S.v[l].ctx));
    S.level_a = on;
}
bool parity_level_a() { return hp_ctx_get_parity_level(engine()) == HP_PARITY_A; }

TransferStats g_stats;

void check(int rc) {
    g_stats.engine_calls++;
    g_stats.calls_by_device[rank_of(lane_set().cur)]++;
    lane_set().v[lane_set().cur].busy = true;
    if (rc == HP_OK) return;
    std::string msg = hp_last_error(cur());   // the calling thread's own last failure (hp_ctx.cpp)
    (void)hp_sync(cur());   // operands may have been enqueued for upload from the caller's memory (limb_copy_h2d): let them finish
                            // before the exception hands that memory back
    if (rc == HP_EINVAL || rc == HP_ERANGE) throw std::invalid_argument(msg);   // (HP_ERANGE: a level-A call was handed a word >= 2 q)
    if (rc == HP_ELOGIC) throw std::logic_error(msg);
    throw std::runtime_error("hehub_amd: " + msg);
}

// printed at exit when HEHUB_AMD_VERBOSE is set, so a run of somebody else's test-suite over this layer can show that the
// work really went to the device and how much crossed PCIe
namespace {
struct Report {
    ~Report() {
        if (std::getenv("HEHUB_AMD_VERBOSE"))
            std::fprintf(stderr, "hehub_amd: %llu engine calls (%s); PCIe: %llu copies / %.1f MiB to the device, %llu copies / %.1f MiB back; "
                                 "%llu host blocks registered for DMA; %llu waits between lanes; %d device rank(s), %llu copies / %.1f MiB between them\n",
                         g_stats.engine_calls, hp_version(), g_stats.h2d_copies, g_stats.h2d_bytes / 1048576.0, g_stats.d2h_copies,
                         g_stats.d2h_bytes / 1048576.0, g_stats.host_blocks_registered, g_stats.lane_waits, lane_set().ndev, g_stats.peer_copies,
                         g_stats.peer_bytes / 1048576.0);
    }
} g_report;
} // namespace
namespace {
// the current lane is ordered behind every call lane l has enqueued so far
void order_after(int l) {
    LaneSet &S = lane_set();
    Lane &me = S.v[S.cur];
    hp_ctx *mine = cur();
    if (hp_ctx_wait_for(mine, S.v[l].ctx) != HP_OK) throw std::runtime_error(std::string("hehub_amd: ") + hp_last_error(mine));
    me.seen[l] = S.v[l].ticket;
    g_stats.lane_waits++;
}

} // namespace

// the current call reads / writes the block: wait for whoever it depends on, leave the call's ticket
void track_read(DevBlock &b0) {
    LaneSet &S = lane_set();
    if (S.count == 1 && S.ndev == 1) return;
    DevBlock &b = b0.parent ? *b0.parent : b0;
    Lane &me = S.v[S.cur];
    for (int l = 0; l < MAX_SLOTS; l++)
        if (l != S.cur && b.wr[l] > me.seen[l]) order_after(l);
    b.rd[S.cur] = me.ticket;
}
void track_write(DevBlock &b0) {
    LaneSet &S = lane_set();
    DevBlock &b = b0.parent ? *b0.parent : b0;
    b.last_wr = S.cur;
    if (S.count == 1 && S.ndev == 1) return;
    Lane &me = S.v[S.cur];
    for (int l = 0; l < MAX_SLOTS; l++)
        if (l != S.cur && std::max(b.wr[l], b.rd[l]) > me.seen[l]) order_after(l);
    b.wr[S.cur] = me.ticket;
}
// the device rank a block's words live on
int home_rank(const DevBlock &b) { return b.parent ? b.parent->rank : b.rank; }
// after a host synchronisation of the current lane that followed track_write: every earlier user of the block has finished
void settled(DevBlock &b) {
    if (b.parent) return;
    for (int l = 0; l < MAX_SLOTS; l++) b.rd[l] = b.wr[l] = 0;
}
TransferStats transfer_stats() { return g_stats; }

} // namespace amd

} // namespace hehub
