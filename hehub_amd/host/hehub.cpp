// hehub.cpp -- implementation of the hehub-compatible host layer over the C ABI.  No arithmetic on
// ring elements happens in this file: it validates arguments the way the reference does, finds (or puts)
// the operands' words on the device, calls the engine and binds the results to the objects it returns.
// Operands stay in HBM between calls: in the own-mirror build every vector carries its device copy
// (hehub.hpp), in the binding build a cache of device copies is keyed by the host objects (opt-in).
//
// Two ways to build it:
//   default                      against hehub.hpp, our own mirror of the reference's types (hehub_amd/host);
//   -DHEHUB_AMD_BIND_REFERENCE   against the reference's OWN headers (-I<hehub>/src): the file then defines
//                                only the functions hehub defines out of line on the hot path (the ones listed in
//                                SURVEY.md section 8a), with hehub's exact signatures, so linking it ahead of
//                                hehub's ntt.cpp / mod_arith.cpp / rns.cpp / rgsw.cpp / rescaling.cpp /
//                                mod_switch.cpp / arith.cpp / permutation.cpp moves that path to the GPU while
//                                everything else (sampling, encoding, key generation, circuits, tests) stays
//                                hehub's.  The `ref_tests` make target does exactly that with hehub's
//                                own test suite.
#ifdef HEHUB_AMD_BIND_REFERENCE
#include "fhe/bgv/bgv.h"
#include "fhe/ckks/ckks.h"
#include "fhe/common/mod_arith.h"
#include "fhe/common/ntt.h"
#include "fhe/common/permutation.h"
#include "fhe/common/rns.h"
#include "fhe/primitives/keys.h"
#include "fhe/primitives/rgsw.h"
#include "fhe/primitives/rlwe.h"
#include "hehub_amd_ext.hpp"
#else
#include "hehub.hpp"
#endif

#include "../../include/hehub_amd.h"

#include <malloc.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

namespace hehub {

namespace amd {

// ---- lanes ------------------------------------------------------------------------------------------------------
// hehub's interface is one ciphertext per call and its callers are loops of INDEPENDENT calls (src/circuits/linear_algebra.h:
// 109-133: two rotations with different keys per diagonal; examples/ckks_example.cpp:15-26; bench/benchmarks.cpp:24-35), and one
// C3 ciphertext fills 10 .. 100 of the 256 CUs.  The own-mirror build therefore spreads the calls over a few LANES of the engine
// (hp_ctx_fork: own stream + scratch, shared tables) and keeps the book on every device block -- which lane wrote it last, which
// lanes read it since -- so that a call waits (on the device: hp_ctx_wait_for, one event) for exactly the calls it depends on:
//   * a call goes to the lane whose most recent call produced one of its operands (a dependent chain stays on one stream and
//     needs no event at all), otherwise to the next lane round robin;
//   * a reader waits for the block's writers on other lanes, a writer (a fresh block out of the pool included) for its readers too;
//   * uploads and downloads are synchronous on their lane, so they leave no debt behind.
// One lane (HEHUB_AMD_LANES=1) is the round-4 behaviour: everything on one stream.  The binding build always has one lane: hehub's
// objects are host memory, every call ends with the download of its result.
//
// DEVICES (round 6).  hehub has no devices; its callers hold many independent ciphertexts (src/circuits/linear_algebra.h:
// 109-133, examples/ckks_example.cpp:10-27), and those "shard naturally across the 8 GPUs" of a node.  With amd::set_devices /
// HEHUB_AMD_DEVICES the layer keeps one engine family (root context + lanes) PER DEVICE RANK; a lane is then a SLOT = (rank, lane).
//   * a device block belongs to the rank it was allocated on; the block pool is kept per rank;
//   * a call runs where its operands live (the rank of its first device-resident operand), a call on host-only operands goes to the
//     next rank round robin and uploads them THERE -- independent ciphertexts spread over the devices, a dependent chain stays put;
//   * recorded calls carry their rank in the signature: a group is one rank's, groups of different ranks overlap;
//   * the batched forms cut a batch into contiguous slices, one per rank (batch / ranks each, no collective);
//   * keys and tables are replicated per rank on first use (the key cache is keyed by rank);
//   * an operand that lives on another rank than the call is copied over (hp_memcpy_peer_async: one xGMI link) and, when it is a
//     vector, stays there -- counted in TransferStats::peer_copies.  Ranks may share a GPU (HEHUB_AMD_DEVICES=0,0: the one-GPU tests).
constexpr int MAX_LANES = 8;                       // lanes per device rank
constexpr int MAX_DEVS = 8;                        // device ranks
constexpr int MAX_SLOTS = MAX_LANES * MAX_DEVS;    // slot = rank * MAX_LANES + lane
inline int rank_of(int slot) { return slot / MAX_LANES; }
struct Lane {
    hp_ctx *ctx = nullptr;
    unsigned long long ticket = 0;            // number of the lane's current / most recent call
    unsigned long long seen[MAX_SLOTS] = {};  // seen[l]: this lane is ordered behind slot l's calls up to that ticket
    bool busy = false;                        // something may still be running on it (cleared by a synchronous copy / a sync on the lane)
};
struct LaneSet {
    Lane v[MAX_SLOTS];
    int count = 1;                            // lanes per rank in use
    int ndev = 1, devs[MAX_DEVS] = {};        // device ranks in use and the HIP device of each
    int cur = 0, rr[MAX_DEVS] = {}, rr_dev = -1, depth = 0, last = -1;   // cur / last: slots; last: the slot of the most recent call (-1: none yet)
    bool level_a = false;
    bool active(int slot) const { return rank_of(slot) < ndev && slot % MAX_LANES < count; }
};
namespace {
LaneSet &lane_set() {
    static LaneSet &S = *new LaneSet;   // (never destroyed: see Pool)
    return S;
}
} // namespace

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
static hp_ctx *cur() {
    hp_ctx *root = engine();
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
        if (!L.ctx) {
            if (hp_ctx_fork(R.ctx, &L.ctx) != HP_OK) throw std::runtime_error(std::string("hehub_amd: ") + hp_last_error(R.ctx));
            (void)hp_ctx_set_parity_level(L.ctx, S.level_a ? HP_PARITY_A : HP_PARITY_B);
        }
        (void)root;
    }
    return L.ctx;
}
static int cur_rank() { return rank_of(lane_set().cur); }
// the root context of a device rank (made on first use)
static hp_ctx *rank_ctx(int rank) {
    LaneSet &S = lane_set();
    if (!S.v[rank * MAX_LANES].ctx) {
        const int saved = S.cur;
        S.cur = rank * MAX_LANES;
        (void)cur();
        S.cur = saved;
    }
    return S.v[rank * MAX_LANES].ctx;
}

int lanes() {
    (void)engine();
    return lane_set().count;
}
void flush_all();
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

} // namespace amd

// =====================================================================================================
// where the words are: pooled device blocks, the two copies of a vector, PCIe accounting
// =====================================================================================================
namespace amd {

// A device allocation out of a per-size free list (hehub pools its host blocks the same way and never gives them back to
// the OS, allocator.h:19-49).  A block carries the tickets of the last call that wrote it and of the last call that read it, per
// lane; the record stays with the block through the pool, so whoever gets it next is ordered behind its previous users.
struct PendingOp;
struct DevBlock {
    u64 *p = nullptr;
    size_t words = 0;
    int rank = 0;      // the device rank the words live on (a placeholder: the rank its recorded call will run on)
    unsigned long long rd[MAX_SLOTS] = {}, wr[MAX_SLOTS] = {};
    int last_wr = 0;   // the slot of the most recent write (tickets are per-lane counters: they do not say which lane wrote LAST)
    // deferred mode (see "deferred execution" below): the result of a call that has been recorded but not run is a PLACEHOLDER
    // (p == NULL, op = the recorded call); when the call runs, the placeholder becomes a view of the block its batch filled
    std::shared_ptr<DevBlock> parent;
    PendingOp *op = nullptr;
    bool failed = false;
    unsigned pending_reads = 0;   // recorded calls that will read this block: an eager write into it has to run them first
};

namespace {

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

// Heap-allocated and never destroyed on purpose: a static object's destructor would run hipFree during static destruction at
// process exit, when the HIP runtime may already be gone (crash or hang at exit).  Pooled blocks go back with the process.
struct Pool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<DevBlock *>> free;   // (rank, words)
    size_t free_bytes = 0;
    size_t cap_bytes = (size_t)8 << 30;   // beyond this a returned block goes back to the device (HEHUB_AMD_POOL_MIB)
};
Pool &pool() {
    static Pool &p = *[] {
        Pool *q = new Pool;
        if (const char *e = std::getenv("HEHUB_AMD_POOL_MIB")) q->cap_bytes = (size_t)std::atol(e) << 20;
        return q;
    }();
    return p;
}

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

using BlockRef = std::shared_ptr<DevBlock>;

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

BlockRef alloc_block(size_t words) {
    if (words == 0) words = 2;
    Pool &P = pool();
    DevBlock *blk = nullptr;
    const int rank = cur_rank();   // (a block is made on the device of the call that asks for it)
    {
        std::lock_guard<std::mutex> lk(P.mu);
        // a pooled block whose previous users are all on the current lane (or have been waited for): taking one that another lane
        // still reads or writes would make this call wait for that lane -- a dependency the program does not have.  (Each lane
        // so ends up recycling its own working set; a block nobody can take yet stays pooled.)
        auto it = P.free.find({rank, words});
        if (it != P.free.end()) {
            LaneSet &S = lane_set();
            const Lane &me = S.v[S.cur];
            auto &list = it->second;
            const bool one = S.count == 1 && S.ndev == 1;
            for (size_t i = list.size(); i-- > 0;) {
                bool clean = true;
                for (int l = 0; l < MAX_SLOTS && clean; l++)
                    clean = l == S.cur || one || std::max(list[i]->wr[l], list[i]->rd[l]) <= me.seen[l];
                if (!clean) continue;
                blk = list[i];
                list.erase(list.begin() + i);
                P.free_bytes -= words * 8;
                break;
            }
        }
    }
    if (!blk) {
        void *d = nullptr;
        check(hp_dev_alloc(cur(), words * sizeof(u64), &d));
        blk = new DevBlock;
        blk->p = (u64 *)d;
        blk->words = words;
        blk->rank = rank;
    }
    return BlockRef(blk, [](DevBlock *b) {
        Pool &Q = pool();
        bool keep;
        {
            std::lock_guard<std::mutex> lk(Q.mu);
            keep = Q.free_bytes + b->words * 8 <= Q.cap_bytes;
            if (keep) {
                Q.free[{b->rank, b->words}].push_back(b);   // (with its record: the next owner waits for this one's readers and writers)
                Q.free_bytes += b->words * 8;
            }
        }
        if (!keep) {
            LaneSet &S = lane_set();
            for (int l = 0; l < MAX_SLOTS; l++)
                if (S.v[l].ctx) (void)hp_sync(S.v[l].ctx);
            (void)hp_dev_free(S.v[b->rank * MAX_LANES].ctx, b->p);
            delete b;
        }
    });
}

// One call of the public interface: picks the lane (see "lanes" above) and opens a new ticket on it.  Calls nest (ckks::add ->
// add -> operator+=): the outermost scope decides.
struct OpScope {
    // force_lane / force_rank >= 0: the call runs on that lane / device rank whatever its operands say (batches and the queue: lane 0 of
    // the rank that holds their operands).  A FORCED scope switches slots also when it is nested inside another call (the queue may run
    // in the middle of a call on another rank; a download runs on the device that holds the words) and switches back when it ends.
    explicit OpScope(std::initializer_list<const BlockRef *> operands, int force_lane = -1, int force_rank = -1) {
        (void)engine();
        LaneSet &S = lane_set();
        saved_ = S.cur;
        const bool forced = force_lane >= 0 || force_rank >= 0;
        if (S.depth++ > 0 && !forced) return;
        const bool nested = S.depth > 1;
        int slot = -1;
        const bool one = S.count == 1 && S.ndev == 1;
        if (one) slot = 0;
        // the lane whose most recent call produced one of the operands: a dependent chain stays on one stream
        if (slot < 0 && !forced)
            for (const BlockRef *r : operands) {
                if (slot >= 0) break;
                if (!r || !*r) continue;
                const DevBlock &b = (*r)->parent ? *(*r)->parent : **r;
                for (int l = 0; l < MAX_SLOTS; l++)
                    if (S.active(l) && b.wr[l] && b.wr[l] == S.v[l].ticket) { slot = l; break; }
            }
        if (slot < 0) {
            // the device: where the first device-resident operand lives; host-only operands go to the next rank round robin
            int rank = force_rank;
            if (rank < 0 && nested) rank = rank_of(S.cur);   // (a forced lane inside another call: that call's rank)
            for (const BlockRef *r : operands) {
                if (rank >= 0) break;
                if (!r || !*r) continue;
                const int h = home_rank(**r);
                if (h < S.ndev) rank = h;
            }
            if (rank < 0) rank = S.ndev == 1 ? 0 : (S.rr_dev = (S.rr_dev + 1) % S.ndev);
            int lane = force_lane;
            if (lane < 0 && S.count == 1) lane = 0;
            if (lane < 0) {
                // nothing in flight on the rank (a caller that looks at every result before its next call, like hehub's own benchmark
                // loop): stay on the lane used last -- its workspace is the one in the Infinity Cache, and there is nothing to overlap with
                bool any_busy = false;
                for (int l = 0; l < S.count; l++) any_busy = any_busy || S.v[rank * MAX_LANES + l].busy;
                const bool stay = !any_busy && S.last >= 0 && rank_of(S.last) == rank;
                lane = stay ? (S.last % MAX_LANES < S.count ? S.last % MAX_LANES : 0) : (S.rr[rank] = (S.rr[rank] + 1) % S.count);
            }
            slot = rank * MAX_LANES + lane;
        }
        S.cur = slot;
        if (!nested) S.last = slot;
        (void)cur();
        S.v[slot].ticket++;
    }
    ~OpScope() {
        LaneSet &S = lane_set();
        if (--S.depth > 0) S.cur = saved_;   // (a forced scope inside another call: that call goes on where it was)
    }
    OpScope(const OpScope &) = delete;

private:
    int saved_ = 0;
};

void h2d(u64 *dst, const u64 *src, size_t words) {
    check(hp_memcpy_h2d(cur(), dst, src, words * sizeof(u64)));
    lane_set().v[lane_set().cur].busy = false;   // (synchronous on its lane)
    g_stats.h2d_bytes += words * 8;
    g_stats.h2d_copies++;
}
void d2h(u64 *dst, const u64 *src, size_t words) {
    check(hp_memcpy_d2h(cur(), dst, src, words * sizeof(u64)));
    lane_set().v[lane_set().cur].busy = false;   // (synchronous on its lane)
    g_stats.d2h_bytes += words * 8;
    g_stats.d2h_copies++;
}

TransferStats transfer_stats() { return g_stats; }

// ---- limbs that live in host memory the CALLER owns (hehub's SmartArray blocks, allocator.h:105-223) ---------------------
// hehub's pool recycles its blocks and never gives one back to the OS (allocator.h:19-22,45-49,204-210), so a block can be
// registered with the driver ONCE (hipHostRegister) and from then on crosses PCIe by DMA at the link rate, asynchronously, instead
// of through the runtime's staging copy of pageable memory (4-5 x slower at 256 KiB).  Only blocks of at least 128 KiB are
// registered: glibc maps those on their own pages, smaller ones share pages with other heap objects.  HEHUB_AMD_PIN_HOST=0
// turns it off.  limb_copy_* only ENQUEUE; limb_copies_wait() is called once per engine call, after the last download.
namespace {
struct PinSet {
    std::mutex mu;
    std::unordered_map<const void *, bool> seen;   // block -> registered (false: the driver refused, do not ask again)
    bool on = true;
};
PinSet &pins() {
    static PinSet &p = *[] {
        PinSet *q = new PinSet;
        if (const char *e = std::getenv("HEHUB_AMD_PIN_HOST")) q->on = std::atoi(e) != 0;
        return q;
    }();
    return p;
}
bool pinned_block(const u64 *p, size_t words) {
#ifndef HEHUB_AMD_BIND_REFERENCE
    // own mirror: limbs are std::vector buffers, which DO go back to the OS when they die -- a registration would outlive the
    // mapping.  (The mirror keeps its words on the device anyway and stages through its own page-locked buffer.)
    (void)p; (void)words;
    return false;
#endif
    PinSet &P = pins();
    if (!P.on || words * sizeof(u64) < (128u << 10)) return false;
    std::lock_guard<std::mutex> lk(P.mu);
    // glibc gives a block of >= M_MMAP_THRESHOLD bytes its own mapping, but RAISES that threshold (up to 32 MiB) whenever such a block
    // is freed, after which later 256 KiB limbs come out of the main heap and share pages with their neighbours.  Setting the
    // threshold explicitly switches the adaptation off: every limb block of 128 KiB and more keeps its own pages for the life of the
    // process.  (A block that nevertheless overlaps registered pages is accepted by hp_host_register when it is fully covered.)
    static const bool fixed_threshold = mallopt(M_MMAP_THRESHOLD, 128 << 10) != 0;
    (void)fixed_threshold;
    auto it = P.seen.find(p);
    if (it != P.seen.end()) return it->second;
    const bool ok = hp_host_register(cur(), const_cast<u64 *>(p), words * sizeof(u64)) == HP_OK;
    if (ok) g_stats.host_blocks_registered++;
    P.seen.emplace(p, ok);
    return ok;
}
} // namespace
void limb_copy_h2d(u64 *dst, const u64 *src, size_t words) {
    if (!pinned_block(src, words)) return h2d(dst, src, words);
    check(hp_memcpy_h2d_async(cur(), dst, src, words * sizeof(u64)));
    g_stats.h2d_bytes += words * 8;
    g_stats.h2d_copies++;
}
void limb_copy_d2h(u64 *dst, const u64 *src, size_t words) {
    if (!pinned_block(dst, words)) return d2h(dst, src, words);
    check(hp_memcpy_d2h_async(cur(), dst, src, words * sizeof(u64)));
    g_stats.d2h_bytes += words * 8;
    g_stats.d2h_copies++;
}
// Limbs that cannot be registered (smaller than 128 KiB: they share pages with other heap objects) cross PCIe through a page-locked
// arena instead, a whole polynomial per DMA: packed by memcpy on the way up, unpacked after the wait on the way down.
// (Arena, the PCIe counters and the lane book are NOT synchronised: like hehub itself -- process-global unsynchronised caches and
// pools, SURVEY.md section 5 -- the layer serves one thread at a time; only the registration set and the block pool take a lock,
// because destructors of objects handed to other threads may run there.)
namespace {
struct Arena {
    u64 *buf = nullptr;
    size_t cap = 0, used = 0;
    struct Pending { std::vector<u64 *> rows; const u64 *st; size_t n; };
    std::vector<Pending> down;
};
Arena &arena() { static Arena &a = *new Arena; return a; }
void arena_flush() {   // everything enqueued so far has happened; hand the downloaded words to their limbs
    Arena &A = arena();
    check(hp_sync(cur()));
    for (auto &p : A.down)
        for (size_t k = 0; k < p.rows.size(); k++) std::memcpy(p.rows[k], p.st + k * p.n, p.n * sizeof(u64));
    A.down.clear();
    A.used = 0;
}
u64 *arena_take(size_t words) {
    Arena &A = arena();
    if (A.used + words > A.cap) {
        arena_flush();
        if (words > A.cap) {
            if (A.buf) (void)hp_host_free(cur(), A.buf);
            void *p = nullptr;
            const size_t want = std::max(words, (size_t)1 << 20);   // at least 8 MiB
            check(hp_host_alloc(cur(), want * sizeof(u64), &p));
            A.buf = (u64 *)p;
            A.cap = want;
        }
    }
    u64 *r = A.buf + A.used;
    A.used += (words + 1) & ~(size_t)1;
    return r;
}
} // namespace
void limb_copies_wait() { arena_flush(); }
// a whole polynomial at once: when every limb is a registered block, ONE kernel moves all of them over PCIe (47-49 GB/s against
// 11-17 GB/s for a DMA command per block); otherwise one DMA through the page-locked arena
template <class Vec> void poly_copy_h2d(u64 *dst, const Vec &v, size_t limbs, size_t n) {
    if (limbs == 0 || n == 0) return;
    bool all = limbs >= 2;
    for (size_t k = 0; k < limbs && all; k++) all = pinned_block(v[(int)k].data(), n);
    if (all) {
        std::vector<const u64 *> rows(limbs);
        for (size_t k = 0; k < limbs; k++) rows[k] = v[(int)k].data();
        check(hp_dev_load_host_rows(cur(), limbs, n, dst, rows.data()));
    } else if (limbs == 1 && pinned_block(v[0].data(), n)) {
        check(hp_memcpy_h2d_async(cur(), dst, v[0].data(), n * sizeof(u64)));
    } else {
        u64 *st = arena_take(limbs * n);
        for (size_t k = 0; k < limbs; k++) std::memcpy(st + k * n, v[(int)k].data(), n * sizeof(u64));
        check(hp_memcpy_h2d_async(cur(), dst, st, limbs * n * sizeof(u64)));
    }
    g_stats.h2d_bytes += limbs * n * 8;
    g_stats.h2d_copies++;
}
template <class Vec> void poly_copy_d2h(Vec &v, const u64 *src, size_t limbs, size_t n) {
    if (limbs == 0 || n == 0) return;
    bool all = limbs >= 2;
    for (size_t k = 0; k < limbs && all; k++) all = pinned_block(v[(int)k].data(), n);
    if (all) {
        std::vector<u64 *> rows(limbs);
        for (size_t k = 0; k < limbs; k++) rows[k] = v[(int)k].data();
        check(hp_dev_store_host_rows(cur(), limbs, n, src, rows.data()));
    } else if (limbs == 1 && pinned_block(v[0].data(), n)) {
        check(hp_memcpy_d2h_async(cur(), v[0].data(), src, n * sizeof(u64)));
    } else {
        u64 *st = arena_take(limbs * n);
        check(hp_memcpy_d2h_async(cur(), st, src, limbs * n * sizeof(u64)));
        Arena::Pending p;
        p.st = st; p.n = n;
        for (size_t k = 0; k < limbs; k++) p.rows.push_back(v[(int)k].data());
        arena().down.push_back(std::move(p));
    }
    g_stats.d2h_bytes += limbs * n * 8;
    g_stats.d2h_copies++;
}

#ifndef HEHUB_AMD_BIND_REFERENCE
// A polynomial's limbs are separate host vectors but one contiguous device view: they cross PCIe as ONE copy through a
// page-locked staging buffer (DMA at the link rate; L pageable copies of 8 N bytes each cost 2-3 x as much at N = 32768).
// The buffer grows to the largest polynomial seen and stays (never freed, like the pool).
namespace {
u64 *pinned(size_t words) {
    static u64 *buf = nullptr;
    static size_t cap = 0;
    if (words > cap) {
        if (buf) (void)hp_host_free(cur(), buf);
        void *p = nullptr;
        check(hp_host_alloc(cur(), words * sizeof(u64), &p));
        buf = (u64 *)p;
        cap = words;
    }
    return buf;
}
} // namespace
void h2d_limbs(u64 *dst, const std::vector<std::vector<u64>> &limbs, size_t count, size_t n) {
    if (count == 0 || n == 0) return;
    if (count == 1) return h2d(dst, limbs[0].data(), n);
    u64 *st = pinned(count * n);
    for (size_t k = 0; k < count; k++) std::memcpy(st + k * n, limbs[k].data(), n * sizeof(u64));
    h2d(dst, st, count * n);
}
void d2h_limbs(std::vector<std::vector<u64>> &limbs, const u64 *src, size_t count, size_t n) {
    if (count == 0 || n == 0) return;
    if (count == 1) return d2h(limbs[0].data(), src, n);
    u64 *st = pinned(count * n);
    d2h(st, src, count * n);
    for (size_t k = 0; k < count; k++) std::memcpy(limbs[k].data(), st + k * n, n * sizeof(u64));
}
#endif

// words on the device for one engine call: `p` points at [polys][limbs][N]; `hold` keeps a temporary / cached block alive
// until the call has been enqueued (the pool hands blocks out in stream order, so that is long enough)
struct Src {
    const u64 *p = nullptr;
    BlockRef hold;
};
// a fresh block for the words an engine call produces
struct Dst {
    BlockRef blk;
    u64 *p = nullptr;
    explicit Dst(size_t words) : blk(alloc_block(words)), p(blk->p) { track_write(*blk); }
};

// ---- deferred execution (own-mirror build; ON by default since round 6, amd::set_deferred(false) / HEHUB_AMD_DEFER=0 turn it off) ----
// (The binding build cannot defer anything: hehub's own objects are host memory the caller may dereference the moment a call returns,
// with no accessor in between -- every call there ends with the download of its result.)
// hehub's interface is one ciphertext per call and its callers loop over INDEPENDENT ciphertexts (src/circuits/linear_algebra.h:
// 109-133, bench/benchmarks.cpp:24-35); at batch 1 a C3 call is a chain of ~12 dependent launches of 40 us that fill 10 .. 100 of
// 256 CUs, and the GPU runs at most 2 - 3 such chains side by side (lanes: x 2.3).  In deferred mode the scheme-level calls
// (mult_low_level, relinearize, rotate / conjugate, rescale_inplace / mod_switch_inplace, add / sub of ciphertexts, polynomial
// products, copies of pending results) are RECORDED, not
// run: every argument check of the single call has been made (same exceptions, same place), the result objects exist and carry
// their shape, scaling factor and a placeholder for their device words.  The queue runs when somebody needs words -- a look at a
// result (operator[], view(), ==), any call that is not deferrable, amd::synchronize(), 1024 recorded calls -- and then groups
// the recorded calls: calls with the same signature (operation, ring degree, moduli, key, step, ...) whose operands are ready run as
// ONE batched engine call (operands gathered by one kernel unless they already lie packed, results views of one block).  A loop of
// 256 independent ckks::mult + rescale_inplace so runs as three batch-256 launches groups, 8 interleaved chains as batch-8 ones.
// Results are word for word those of the eager calls.  What differs: a failure INSIDE the engine (a HIP error, a modulus the
// transforms reject) surfaces when the queue runs, not at the call that recorded it.
enum class OpKind { MultLow, Relin, KeySwitch, Drop, AddSub, Copy, PolyMul, Transform, PolyAddSub, BaseConv };
struct PendingOp {
    OpKind kind = OpKind::MultLow;
    size_t logn = 0, L = 0, L0 = 0, step = 0;
    int rank = 0;           // the device rank the call runs on: where its first device-resident operand lived when it was recorded
    std::vector<u64> mod;   // q_0 .. q_{L-1} (MultLow, Drop, AddSub) or the extended chain (Relin, KeySwitch)
    BlockRef key;           // the assembled key block (Relin, KeySwitch)
    bool conj = false, bgv = false, sub = false;
    u64 t = 0;
    std::vector<std::pair<BlockRef, size_t>> in;   // operand polynomials: block + word offset; `in_limbs` limbs are read of each
    size_t in_limbs = 0, out_words = 0;
    BlockRef out;           // placeholder of out_words words
    bool done = false;
    // (rotations / conjugations group ACROSS keys and steps: the engine takes a key and a step per ciphertext, hp_dev_ckks_rotate_many --
    // the rotations of one vector under the keys of a rotation key set, src/circuits/linear_algebra.h:123-130, are one launch sequence)
    bool same_signature(const PendingOp &o) const {
        const bool ks = kind == OpKind::KeySwitch;
        return kind == o.kind && rank == o.rank && logn == o.logn && L == o.L && L0 == o.L0 && (ks || (step == o.step && conj == o.conj && key == o.key)) &&
               bgv == o.bgv && sub == o.sub && t == o.t && in_limbs == o.in_limbs && in.size() == o.in.size() && mod == o.mod;
    }
    bool ready() const {
        for (auto &r : in)
            if (r.first->op) return false;
        return true;
    }
};
struct OpQueue {
    std::vector<std::unique_ptr<PendingOp>> ops;
    bool on = false, flushing = false;
    static constexpr size_t MAX_PENDING = 1024;
};
namespace {
OpQueue &op_queue() {
    static OpQueue &q = *[] {
        OpQueue *x = new OpQueue;
#ifndef HEHUB_AMD_BIND_REFERENCE
        x->on = true;   // (the default since round 6: an unchanged loop of single calls gets the batch rate; HEHUB_AMD_DEFER=0 is the escape)
        if (const char *e = std::getenv("HEHUB_AMD_DEFER")) x->on = std::atoi(e) != 0;
#endif
        return x;
    }();
    return q;
}
} // namespace
void flush_all();
BlockRef record(std::unique_ptr<PendingOp> op);
// device address of a block's words; a placeholder is resolved by running the queue
u64 *words_of(const BlockRef &b) {
    if (b->op) flush_all();
    if (!b->p) throw std::runtime_error("hehub_amd: this object is the result of a deferred call that failed when the queue ran");
    return b->p;
}
// `words` words at [off, ..) of a block, enqueued for copying from the device rank that holds them into `dst` on the CURRENT rank
// (ordered behind the block's writers; the block records the read)
void peer_fetch(u64 *dst, const BlockRef &b, size_t off, size_t words) {
    const u64 *src = words_of(b) + off;
    track_read(*b);
    check(hp_memcpy_peer_async(cur(), dst, rank_ctx(home_rank(*b)), src, words * sizeof(u64)));
    g_stats.peer_copies++;
    g_stats.peer_bytes += words * 8;
}
// the words [off, off + words) of a block for an engine call on the CURRENT rank: where they are when they live on this rank, otherwise
// a copy made here for this call (a recorded operand has no vector to re-home; Access::in moves a vector for good)
Src here(const BlockRef &b, size_t off, size_t words) {
    u64 *p = words_of(b);
    if (home_rank(*b) == cur_rank()) {
        track_read(*b);
        return Src{p + off, b};
    }
    BlockRef tmp = alloc_block(words);
    track_write(*tmp);
    peer_fetch(tmp->p, b, off, words);
    return Src{tmp->p, tmp};
}
// a key block assembled for another rank than the call's is a bug of this file, not of the caller (keys are cached per rank)
const u64 *key_here(const BlockRef &key) {
    if (home_rank(*key) != cur_rank()) throw std::logic_error("hehub_amd: key block of another device rank (internal error)");
    track_read(*key);
    return key->p;
}

#ifndef HEHUB_AMD_BIND_REFERENCE
// ---- own mirror: the vector carries its device copy -----------------------------------------------------------
namespace {
unsigned long long next_stamp() {
    static unsigned long long s = 0;
    return ++s;
}
} // namespace

struct Access {
    static size_t words(const RnsIntVec &v, size_t limbs) { return limbs * v.dimension(); }
    // the device copy of the first `limbs` limbs, uploading the host words if they are newer
    static Src in(const RnsIntVec &v, size_t limbs) {
        const size_t n = v.dimension();
        if (v.dev_ok_ && v.blk_->op) flush_all();   // a placeholder: the recorded calls run now
        if (!v.dev_ok_) {
            if (v.blk_ && (v.blk_->pending_reads || (v.blk_->parent && v.blk_->parent->pending_reads))) flush_all();   // (a recorded call still wants the words this upload replaces)
            // (a view keeps its place: sibling views are disjoint -- unless that place is on another device than this call)
            if (!v.blk_ || v.off_ + v.count_ * n > v.blk_->words || home_rank(*v.blk_) != cur_rank()) {
                v.blk_ = alloc_block(v.count_ * n);
                v.off_ = 0;
            }
            const bool whole = v.off_ == 0 && v.count_ * n == v.blk_->words;
            track_write(*v.blk_);
            h2d_limbs(words_of(v.blk_) + v.off_, v.limbs_, v.count_, n);   // (synchronous: the upload leaves no debt on its lane)
            if (whole) settled(*v.blk_);
            v.dev_ok_ = true;
        }
        (void)limbs;
        if (home_rank(*v.blk_) != cur_rank()) move_here(v);
        track_read(*v.blk_);
        return Src{words_of(v.blk_) + v.off_, v.blk_};
    }
    // the vector's device words live on another rank than the current call: they are copied over (one peer copy, ordered behind their
    // writers) and the vector lives HERE from now on -- same words, another place: invisible to the caller
    static void move_here(const RnsIntVec &v) {
        const size_t w = v.count_ * v.dimension();
        BlockRef nb = alloc_block(w);
        track_write(*nb);
        peer_fetch(nb->p, v.blk_, v.off_, w);
        v.blk_ = nb;
        v.off_ = 0;
    }
    // deferred mode: where the vector's device words are or WILL be (a placeholder is not resolved); host words are uploaded now
    static std::pair<BlockRef, size_t> ref(const RnsIntVec &v) {
        if (!v.dev_ok_) (void)in(v, v.count_);
        return {v.blk_, v.off_};
    }
    // a result polynomial whose words are the view [off, ..) of a block (or placeholder)
    static void bind_block(RnsIntVec &v, const BlockRef &blk, size_t off) {
        v.blk_ = blk;
        v.off_ = off;
        v.dev_ok_ = true;
        v.host_ok_ = false;
        v.stamp_ = next_stamp();
    }
    // the block that holds the vector's current device words, if any: what OpScope looks at to keep a dependent chain on one lane
    static const BlockRef *home(const RnsIntVec &v) { return v.dev_ok_ ? &v.blk_ : nullptr; }
    // the vector's own device words, to be overwritten in place by an engine call that has read them (operator+= ...)
    static u64 *inout(RnsIntVec &v) {
        // (a recorded call may read the words this call overwrites: it runs first; a vector no recorded call knows is simply written)
        if (v.blk_ && (v.blk_->op || v.blk_->pending_reads || (v.blk_->parent && v.blk_->parent->pending_reads))) flush_all();
        Src s = in(v, v.count_);
        track_write(*v.blk_);
        v.host_ok_ = false;
        v.stamp_ = next_stamp();
        return const_cast<u64 *>(s.p);
    }
    // a result vector of the given shape whose words are the view [off, off + limbs * N) of a block an engine call fills
    static void shape(RnsIntVec &v, size_t n, size_t limbs, const std::vector<u64> &moduli) {
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        v.logn_ = lg;
        v.count_ = limbs;
        v.q_.assign(moduli.begin(), moduli.begin() + limbs);
        v.limbs_.clear();
    }
    static void bind(RnsIntVec &v, const Dst &d, size_t off, size_t limbs) {
        (void)limbs;
        v.blk_ = d.blk;
        v.off_ = off;
        v.dev_ok_ = true;
        v.host_ok_ = false;   // (the host vectors, if any, stay allocated: a reference a caller still holds reads stale words, not freed memory)
        v.stamp_ = next_stamp();
    }
    // identity of the words for the key cache: exact (every way to change the words changes the stamp)
    static unsigned long long stamp(const RnsIntVec &v) {
        if (!v.stamp_) v.stamp_ = next_stamp();
        return v.stamp_;
    }
    static bool adjacent(const RnsIntVec &a, const RnsIntVec &b, size_t limbs) {
        return a.dev_ok_ && b.dev_ok_ && a.blk_ == b.blk_ && b.off_ == a.off_ + limbs * a.dimension() && home_rank(*a.blk_) == cur_rank();
    }
    // give the device copy up when the host copy is current too (the block returns to the pool once its last user is gone)
    static void drop_device_copy(const RnsIntVec &v) {
        if (!v.host_ok_ || !v.dev_ok_) return;
        v.dev_ok_ = false;
        v.blk_.reset();
        v.off_ = 0;
    }
    // move the (current) device copy to another place that already holds the same words
    static void rehome(const RnsIntVec &v, const BlockRef &blk, size_t off) {
        if (!v.dev_ok_) return;
        v.blk_ = blk;
        v.off_ = off;
    }
    static void sync_host(const RnsIntVec &v) {
        if (v.host_ok_) return;
        const size_t n = v.dimension();
        v.limbs_.resize(v.count_);   // (vectors that exist are refreshed in place: a reference a caller holds sees the new words)
        for (size_t k = 0; k < v.count_; k++) v.limbs_[k].resize(n);
        {
            // the download runs on the lane that wrote the words last (no event needed there), behind the block's other writers
            LaneSet &S = lane_set();
            if (v.blk_->op) flush_all();
            const DevBlock &root = v.blk_->parent ? *v.blk_->parent : *v.blk_;
            (void)S;
            const int slot = S.active(root.last_wr) && rank_of(root.last_wr) == root.rank ? root.last_wr : root.rank * MAX_LANES;
            OpScope op({}, slot % MAX_LANES, rank_of(slot));   // (on the device that holds the words, whatever call this look is part of)
            track_read(*v.blk_);
            d2h_limbs(v.limbs_, words_of(v.blk_) + v.off_, v.count_, n);   // (synchronous)
        }
        v.host_ok_ = true;
    }
    // limb k alone: a caller that looks at one word of a result (`ct[1][0][0]`) pays the PCIe time of one limb, not of the polynomial
    static void sync_host_limb(const RnsIntVec &v, size_t k) {
        if (v.host_ok_) return;
        if (k >= v.count_ || v.count_ > 64) { sync_host(v); return; }
        if (v.mask_stamp_ != stamp(v)) { v.limb_mask_ = 0; v.mask_stamp_ = v.stamp_; }
        const size_t n = v.dimension();
        if (v.limbs_.size() != v.count_) v.limbs_.resize(v.count_);
        if ((v.limb_mask_ >> k) & 1ull && v.limbs_[k].size() == n) return;
        // a second limb is asked for: the caller is walking through the vector -- the rest comes down as ONE copy
        if (v.limb_mask_ != 0) { sync_host(v); return; }
        {
            v.limbs_[k].resize(n);
            LaneSet &S = lane_set();
            if (v.blk_->op) flush_all();
            const DevBlock &root = v.blk_->parent ? *v.blk_->parent : *v.blk_;
            (void)S;
            const int slot = S.active(root.last_wr) && rank_of(root.last_wr) == root.rank ? root.last_wr : root.rank * MAX_LANES;
            OpScope op({}, slot % MAX_LANES, rank_of(slot));
            track_read(*v.blk_);
            d2h(v.limbs_[k].data(), words_of(v.blk_) + v.off_ + k * n, n);   // (synchronous)
            v.limb_mask_ |= 1ull << k;
        }
        if (v.limb_mask_ == (v.count_ == 64 ? ~0ull : ((1ull << v.count_) - 1ull))) v.host_ok_ = true;   // every limb has come down by now
    }
    static void host_written(RnsIntVec &v) {
        sync_host(v);
        if (v.dev_ok_) g_stats.device_copies_invalidated++;   // a non-const access: the next engine call uploads the vector again
        v.dev_ok_ = false;
        v.stamp_ = next_stamp();
    }
    static void copy_from(RnsIntVec &dst, const RnsIntVec &o) {
        dst.logn_ = o.logn_; dst.count_ = o.count_; dst.q_ = o.q_;
        dst.blk_.reset(); dst.off_ = 0; dst.limbs_.clear();
        dst.stamp_ = next_stamp();
        if (o.dev_ok_ && o.count_) {   // device-to-device: the host copy (if any) is not duplicated, it can be fetched again
            const size_t w = o.count_ * o.dimension();
            if (o.blk_->op && deferred() && (w & 1) == 0) {   // a copy of a result that is still a placeholder is recorded like the call that makes it
                std::unique_ptr<PendingOp> rec(new PendingOp);
                rec->kind = OpKind::Copy; rec->logn = o.logn_; rec->L = o.count_; rec->in_limbs = o.count_; rec->out_words = w;
                rec->rank = home_rank(*o.blk_);
                rec->in.push_back({o.blk_, o.off_});
                dst.blk_ = record(std::move(rec));
                dst.dev_ok_ = true;
                dst.host_ok_ = false;
                return;
            }
            if (o.blk_->op) flush_all();   // (a copy of a placeholder: the recorded calls run now)
            OpScope op({&o.blk_});
            dst.blk_ = alloc_block(w);
            track_write(*dst.blk_);
            if (home_rank(*o.blk_) == cur_rank()) {
                track_read(*o.blk_);
                check(hp_dev_copy(cur(), w, words_of(o.blk_) + o.off_, dst.blk_->p));
            } else {
                peer_fetch(dst.blk_->p, o.blk_, o.off_, w);   // (the source's rank went out of use: the copy is made on a rank that is)
            }
            dst.dev_ok_ = true;
            dst.host_ok_ = false;
        } else {
            dst.limbs_ = o.limbs_;
            dst.host_ok_ = true;
            dst.dev_ok_ = false;
        }
    }
    // u64[polys.size()][limbs][N] for a batch entry point: the polynomials' own words when they already lie like that (the result
    // of an earlier batched call, untouched since), otherwise ONE gather kernel into a block that then becomes their home
    static Src batch_in(const std::vector<const RnsIntVec *> &polys, size_t limbs) {
        const RnsIntVec &f = *polys[0];
        const size_t w = limbs * f.dimension();
        bool packed = true;
        for (size_t r = 0; r < polys.size() && packed; r++) {
            const RnsIntVec &v = *polys[r];
            packed = v.dev_ok_ && !v.blk_->op && v.blk_->p && f.blk_->p && v.blk_->p + v.off_ == f.blk_->p + f.off_ + r * w && v.count_ == limbs &&
                     home_rank(*v.blk_) == cur_rank();
        }
        if (packed) {   // (polynomials that are views of one block, directly or through the placeholders a deferred batch resolved)
            for (const RnsIntVec *v : polys) track_read(*v->blk_);
            return Src{f.blk_->p + f.off_, f.blk_->parent ? f.blk_->parent : f.blk_};
        }
        std::vector<const u64 *> rows(polys.size());
        std::vector<BlockRef> holds(polys.size());
        for (size_t r = 0; r < polys.size(); r++) {
            Src s = in(*polys[r], limbs);
            rows[r] = s.p;
            holds[r] = s.hold;
        }
        BlockRef tmp = alloc_block(w * polys.size());
        track_write(*tmp);
        check(hp_dev_gather_rows(cur(), polys.size(), w, rows.data(), tmp->p));
        for (size_t r = 0; r < polys.size(); r++)
            if (polys[r]->count_ == limbs) rehome(*polys[r], tmp, r * w);   // (same words, another place: invisible to the caller)
        return Src{tmp->p, tmp};
    }
    // polynomial r of a batched result is the view [r * limbs * N, (r + 1) * limbs * N) of the block the engine call filled
    static void bind_many(const std::vector<RnsIntVec *> &polys, const Dst &d, size_t limbs) {
        for (size_t r = 0; r < polys.size(); r++) bind(*polys[r], d, r * limbs * polys[r]->dimension(), limbs);
    }
    static void steal(RnsIntVec &dst, RnsIntVec &o) {
        dst.logn_ = o.logn_; dst.count_ = o.count_; dst.q_ = std::move(o.q_); dst.limbs_ = std::move(o.limbs_);
        dst.blk_ = std::move(o.blk_); dst.off_ = o.off_; dst.host_ok_ = o.host_ok_; dst.dev_ok_ = o.dev_ok_; dst.stamp_ = o.stamp_;
        o.logn_ = 0; o.count_ = 0; o.q_.clear(); o.limbs_.clear(); o.blk_.reset(); o.off_ = 0; o.host_ok_ = true; o.dev_ok_ = false;
        o.stamp_ = 0;
    }
};

#else
// ---- binding hehub's own types: host objects are hehub's, the device side is a cache --------------------------------
// hehub's limbs are host memory that anybody may read at any time, so every result is copied back when it is produced.  What
// can be saved is the way TO the device: with HEHUB_AMD_CT_CACHE=<entries> the layer remembers which device block holds the
// words of which host polynomial (recognised by the address of its first limb, its shape and four sampled words of every
// limb) -- an operand that was uploaded or produced by an earlier call is then not uploaded again.  Like the key cache below
// it is opt-in: a polynomial that is modified in place on the host without touching any sampled word would go unnoticed.
namespace {

struct CtCache {
    struct Entry {
        std::vector<u64> sig;   // address, limbs, n, then 4 words per limb
        BlockRef blk;
        size_t off = 0;
    };
    std::mutex mu;
    std::vector<Entry> lru;   // most recently used last
    size_t cap = 0;
};
CtCache &ct_cache() {
    static CtCache &c = *[] {
        CtCache *q = new CtCache;
        if (const char *e = std::getenv("HEHUB_AMD_CT_CACHE")) q->cap = (size_t)std::atol(e);
        return q;
    }();
    return c;
}
std::vector<u64> signature(const RnsIntVec &v, size_t limbs) {
    const size_t n = v.dimension();
    std::vector<u64> sig{(u64)(uintptr_t)v[0].data(), (u64)limbs, (u64)n};
    for (size_t k = 0; k < limbs; k++) {
        const u64 *w = v[(int)k].data();
        sig.insert(sig.end(), {w[0], w[n / 3], w[(2 * n) / 3], w[n - 1]});
    }
    return sig;
}
void cache_put(const RnsIntVec &v, size_t limbs, const BlockRef &blk, size_t off) {
    CtCache &C = ct_cache();
    if (!C.cap || limbs == 0) return;
    auto sig = signature(v, limbs);
    std::lock_guard<std::mutex> lk(C.mu);
    for (size_t i = 0; i < C.lru.size(); i++)
        if (C.lru[i].sig[0] == sig[0]) {   // one entry per host address
            C.lru.erase(C.lru.begin() + i);
            break;
        }
    if (C.lru.size() >= C.cap) C.lru.erase(C.lru.begin());
    C.lru.push_back(CtCache::Entry{std::move(sig), blk, off});
}
bool cache_get(const RnsIntVec &v, size_t limbs, BlockRef &blk, size_t &off) {
    CtCache &C = ct_cache();
    if (!C.cap || limbs == 0) return false;
    const u64 addr = (u64)(uintptr_t)v[0].data();
    std::lock_guard<std::mutex> lk(C.mu);
    for (size_t i = 0; i < C.lru.size(); i++) {
        auto &e = C.lru[i];
        if (e.sig[0] != addr) continue;
        // an entry made for MORE limbs serves a prefix (ciphertext after remove_components): compare the common part
        if (e.sig[2] != v.dimension() || e.sig[1] < limbs) return false;
        auto sig = signature(v, limbs);
        for (size_t j = 3; j < sig.size(); j++)
            if (sig[j] != e.sig[j]) return false;
        auto hit = e;
        C.lru.erase(C.lru.begin() + i);
        C.lru.push_back(hit);
        blk = hit.blk;
        off = hit.off;
        return true;
    }
    return false;
}

} // namespace

struct Access {
    static size_t words(const RnsIntVec &v, size_t limbs) { return limbs * v.dimension(); }
    static Src in(const RnsIntVec &v, size_t limbs) {
        BlockRef blk;
        size_t off = 0;
        if (cache_get(v, limbs, blk, off)) return Src{blk->p + off, blk};
        const size_t n = v.dimension();
        blk = alloc_block(limbs * n);
        // (enqueued; the call that consumes the block ends with the download of its result and limb_copies_wait())
        poly_copy_h2d(blk->p, v, limbs, n);
        cache_put(v, limbs, blk, 0);
        return Src{blk->p, blk};
    }
    static const BlockRef *home(const RnsIntVec &) { return nullptr; }   // (one lane in this build)
    static void shape(RnsIntVec &v, size_t n, size_t limbs, const std::vector<u64> &moduli) {
        v = RnsIntVec(RnsIntVec::Params{n, limbs, std::vector<u64>(moduli.begin(), moduli.begin() + limbs)});
    }
    // hehub's object is host memory: the result comes back now; the device copy is remembered for the next consumer.
    // Several polynomials of one result (the halves of a ciphertext) are enqueued together and waited for once.
    static void bind_enqueue(RnsIntVec &v, const Dst &d, size_t off, size_t limbs) { poly_copy_d2h(v, d.p + off, limbs, v.dimension()); }
    static void bind_finish(RnsIntVec &v, const Dst &d, size_t off, size_t limbs) { cache_put(v, limbs, d.blk, off); }
    static void bind(RnsIntVec &v, const Dst &d, size_t off, size_t limbs) {
        bind_enqueue(v, d, off, limbs);
        limb_copies_wait();   // hehub's object is host memory the caller may read as soon as we return
        bind_finish(v, d, off, limbs);
    }
    // a batch goes up into ONE block, polynomial by polynomial (each a single kernel over PCIe from its registered limb blocks, or
    // one DMA through the page-locked arena); a polynomial the ciphertext cache knows is copied on the device instead
    static Src batch_in(const std::vector<const RnsIntVec *> &polys, size_t limbs) {
        const size_t n = polys[0]->dimension(), w = limbs * n;
        BlockRef tmp = alloc_block(w * polys.size());
        for (size_t r = 0; r < polys.size(); r++) {
            BlockRef blk;
            size_t off = 0;
            if (cache_get(*polys[r], limbs, blk, off)) {
                check(hp_dev_copy(cur(), w, blk->p + off, tmp->p + r * w));
            } else {
                poly_copy_h2d(tmp->p + r * w, *polys[r], limbs, n);
                cache_put(*polys[r], limbs, tmp, r * w);
            }
        }
        return Src{tmp->p, tmp};
    }
    // hehub's objects are host memory: the whole batch is enqueued for download and waited for once
    static void bind_many(const std::vector<RnsIntVec *> &polys, const Dst &d, size_t limbs) {
        for (size_t r = 0; r < polys.size(); r++) bind_enqueue(*polys[r], d, r * limbs * polys[r]->dimension(), limbs);
        limb_copies_wait();
        for (size_t r = 0; r < polys.size(); r++) bind_finish(*polys[r], d, r * limbs * polys[r]->dimension(), limbs);
    }
    static bool adjacent(const RnsIntVec &a, const RnsIntVec &b, size_t limbs) {
        BlockRef ba, bb;
        size_t oa = 0, ob = 0;
        return cache_get(a, limbs, ba, oa) && cache_get(b, limbs, bb, ob) && ba == bb && ob == oa + limbs * a.dimension();
    }
};
#endif

// [polys.size()][limbs][N] contiguous on the device: the polynomials' own words when they already lie like that (the two
// halves of a ciphertext an engine call produced), otherwise gathered into a temporary block by device copies
Src gather(std::initializer_list<const RnsIntVec *> polys, size_t limbs) {
    const RnsIntVec *first = *polys.begin();
    bool adj = true;
    const RnsIntVec *prev = nullptr;
    for (const RnsIntVec *p : polys) {
        if (prev && !Access::adjacent(*prev, *p, limbs)) adj = false;
        prev = p;
    }
    if (polys.size() == 1 || adj) return Access::in(*first, limbs);
    const size_t w = Access::words(*first, limbs);
    BlockRef tmp = alloc_block(w * polys.size());
    track_write(*tmp);
    size_t i = 0;
    bool whole = true;
    for (const RnsIntVec *p : polys) {
        Src s = Access::in(*p, limbs);
        if (w) check(hp_dev_copy(cur(), w, s.p, tmp->p + i * w));
        whole = whole && p->component_count() == limbs;
        i++;
    }
#ifndef HEHUB_AMD_BIND_REFERENCE
    // the gathered block becomes the polynomials' home (same words, another place: invisible to the caller), so the next
    // call finds the halves of this ciphertext side by side and copies nothing
    if (whole) {
        i = 0;
        for (const RnsIntVec *p : polys) Access::rehome(*p, tmp, (i++) * w);
    }
#endif
    return Src{tmp->p, tmp};
}


// ---- deferred execution: running the queue -----------------------------------------------------------------------------------
bool deferred() { return op_queue().on; }

namespace {

// a call has run: it lets go of its operands at once (a dependent chain recycles its blocks through the pool while the queue
// runs, like the eager calls do, instead of holding every intermediate result until the end)
void release_operands(PendingOp &o) {
    for (auto &r : o.in) {
        if (r.first->pending_reads) r.first->pending_reads--;
        if (r.first->parent && r.first->parent->pending_reads) r.first->parent->pending_reads--;
    }
    o.in.clear();
}

// operand polynomials [first, first + count) of every call of a group as u64[B][count][in_limbs][N]: their own words when they
// already lie like that, otherwise one gather kernel
Src group_rows(const std::vector<PendingOp *> &g, size_t first, size_t count, size_t n) {
    const size_t w = g[0]->in_limbs * n;
    const u64 *base = words_of(g[0]->in[first].first) + g[0]->in[first].second;
    bool packed = true;
    std::vector<const u64 *> rows;
    std::vector<BlockRef> holds;   // (copies of operands that live on another rank: alive until the gather has been enqueued)
    rows.reserve(g.size() * count);
    for (size_t b = 0; b < g.size(); b++)
        for (size_t c = 0; c < count; c++) {
            const auto &r = g[b]->in[first + c];
            Src s = here(r.first, r.second, w);
            if (s.hold != r.first) holds.push_back(s.hold);
            packed = packed && s.p == base + (b * count + c) * w && holds.empty();
            rows.push_back(s.p);
        }
    if (packed) return Src{base, nullptr};   // (the calls of the group hold their operand blocks until the group has been enqueued)
    BlockRef tmp = alloc_block(rows.size() * w);
    track_write(*tmp);
    check(hp_dev_gather_rows(cur(), rows.size(), w, rows.data(), tmp->p));
    return Src{tmp->p, tmp};
}

void run_group(const std::vector<PendingOp *> &g) {
    const PendingOp &o = *g[0];
    const size_t B = g.size(), n = (size_t)1 << o.logn, L = o.L;
    BlockRef big = alloc_block(B * o.out_words);
    track_write(*big);
    switch (o.kind) {
    case OpKind::MultLow: {
        Src d1 = group_rows(g, 0, 2, n), d2 = group_rows(g, 2, 2, n);
        check(hp_dev_mult_low_level(cur(), o.logn, L, o.mod.data(), B, d1.p, d2.p, big->p));
        break;
    }
    case OpKind::Relin: {
        Src dq = group_rows(g, 0, 3, n);
        const u64 *key = key_here(o.key);
        if (o.bgv) check(hp_dev_bgv_relinearize(cur(), o.logn, L, o.mod.data(), 1 /* bgv.h:32 */, B, dq.p, key, big->p));
        else check(hp_dev_ckks_relinearize_at(cur(), o.logn, L, o.L0, o.mod.data(), B, dq.p, key, big->p));
        break;
    }
    case OpKind::KeySwitch: {
        bool one_key = true;
        for (PendingOp *c : g) {
            (void)key_here(c->key);
            one_key = one_key && c->key == o.key && c->step == o.step && c->conj == o.conj;
        }
        if (!one_key) {   // every ciphertext with its own key and step; the operands are read where they are (often ONE vector)
            std::vector<const u64 *> keys, polys;
            std::vector<size_t> steps;
            std::vector<unsigned char> conj;
            std::vector<Src> holds;
            for (PendingOp *c : g) {
                keys.push_back(c->key->p);
                steps.push_back(c->step);
                conj.push_back(c->conj ? 1 : 0);
                for (size_t h = 0; h < 2; h++) {
                    holds.push_back(here(c->in[h].first, c->in[h].second, L * n));
                    polys.push_back(holds.back().p);
                }
            }
            check(hp_dev_ckks_rotate_many_rows(cur(), o.logn, L, o.L0, o.mod.data(), B, steps.data(), conj.data(), polys.data(), keys.data(), big->p));
            g_stats.deferred_many_key_groups++;
            break;
        }
        Src dc = group_rows(g, 0, 2, n);
        if (o.conj) check(hp_dev_ckks_conjugate_at(cur(), o.logn, L, o.L0, o.mod.data(), B, dc.p, key_here(o.key), big->p));
        else check(hp_dev_ckks_rotate_at(cur(), o.logn, L, o.L0, o.mod.data(), B, o.step, dc.p, key_here(o.key), big->p));
        break;
    }
    case OpKind::Drop: {
        Src dc = group_rows(g, 0, 2, n);
        if (o.bgv) check(hp_dev_bgv_mod_switch(cur(), o.logn, L, o.mod.data(), o.t, B, dc.p, big->p));
        else check(hp_dev_ckks_rescale(cur(), o.logn, L, o.mod.data(), B, dc.p, big->p));
        break;
    }
    case OpKind::Copy: {   // a deep copy of a result that has not been computed yet (`ct_sum = ct_prod`): the gather IS the copy
        std::vector<const u64 *> rows;
        std::vector<Src> holds;
        for (PendingOp *c : g) {
            holds.push_back(here(c->in[0].first, c->in[0].second, o.in_limbs * n));
            rows.push_back(holds.back().p);
        }
        check(hp_dev_gather_rows(cur(), rows.size(), o.in_limbs * n, rows.data(), big->p));
        break;
    }
    case OpKind::Transform: {   // NTT / INTT of a polynomial in place (ntt.h:41-92): the plaintext transforms inside add / sub / mult_plain (ckks/arith.cpp:25,41,49)
        std::vector<const u64 *> rows;   // (the operands' blocks may have other holders: the batch is transformed in its own block)
        std::vector<Src> holds;
        for (PendingOp *c : g) {
            holds.push_back(here(c->in[0].first, c->in[0].second, o.in_limbs * n));
            rows.push_back(holds.back().p);
        }
        check(hp_dev_gather_rows(cur(), rows.size(), o.in_limbs * n, rows.data(), big->p));
        if (o.conj) check(hp_dev_intt(cur(), o.logn, L, o.mod.data(), B, big->p, o.sub ? 1 : 0));
        else check(hp_dev_ntt(cur(), o.logn, L, o.mod.data(), B, big->p));
        break;
    }
    case OpKind::PolyMul: {   // operator* of two polynomials (rns.cpp:120-140): the plaintext products of mult_plain
        Src da = group_rows(g, 0, 1, n), db = group_rows(g, 1, 1, n);
        check(hp_dev_poly_mul(cur(), n, L, o.mod.data(), B, da.p, db.p, big->p));
        break;
    }
    case OpKind::PolyAddSub: {   // += / -= of two polynomials (rns.cpp:59-98): the plaintext sums of add_plain / sub_plain, the halves of a sum taken apart
        Src da = group_rows(g, 0, 1, n), db = group_rows(g, 1, 1, n);
        if (o.sub) check(hp_dev_poly_sub(cur(), n, L, o.mod.data(), B, da.p, db.p, big->p));
        else check(hp_dev_poly_add(cur(), n, L, o.mod.data(), B, da.p, db.p, big->p));
        break;
    }
    case OpKind::BaseConv: {   // rns_base_transform, one modulus (t) -> many (rns_transform.cpp:113 + :11-37): the plaintext lift of the bgv plain operations
        Src din = group_rows(g, 0, 1, n);
        check(hp_dev_rns_base_from_single(cur(), n, o.t, L, o.mod.data(), B, din.p, big->p));
        break;
    }
    case OpKind::AddSub: {
        Src da = group_rows(g, 0, 2, n), db = group_rows(g, 2, 2, n);
        if (o.sub) check(hp_dev_poly_sub(cur(), n, L, o.mod.data(), 2 * B, da.p, db.p, big->p));
        else check(hp_dev_poly_add(cur(), n, L, o.mod.data(), 2 * B, da.p, db.p, big->p));
        break;
    }
    }
    for (size_t b = 0; b < B; b++) {   // the placeholders become views of the block the batch filled
        DevBlock &ph = *g[b]->out;
        ph.p = big->p + b * o.out_words;
        ph.parent = big;
        ph.op = nullptr;
        g[b]->done = true;
        release_operands(*g[b]);
    }
    g_stats.deferred_groups++;
    g_stats.deferred_calls += B;
}

// The fused pipeline: a group of mult_low_level calls every one of which feeds exactly one recorded relinearize whose result feeds
// exactly one recorded rescale_inplace / mod_switch_inplace, with NOBODY else holding the intermediate results (the tensor product
// of an inline ckks::mult dies inside it, the relinearised ciphertext was rebound by the in-place drop): that is ckks::mult +
// rescale_inplace in a loop, and it runs as the engine's one-call pipeline (hp_dev_ckks_mult_relin_rescale: no quadratic or
// intermediate ciphertexts in HBM, at level A the two drops as one transform) -- same words as the three separate calls.
// Returns the calls of the group that were NOT part of such a triple (they run as an ordinary group).
std::vector<PendingOp *> run_fused_mults(const std::vector<std::unique_ptr<PendingOp>> &ops, const std::vector<PendingOp *> &g_all) {
    if (g_all[0]->kind != OpKind::MultLow) return g_all;
    std::vector<PendingOp *> g, rest;
    const size_t n = (size_t)1 << g_all[0]->logn, L = g_all[0]->L, w = L * n;
    std::vector<PendingOp *> relin, drop;
    auto consumer_of = [&](const BlockRef &ph, OpKind kind, size_t polys) -> PendingOp * {
        if ((size_t)ph.use_count() != 1 + polys) return nullptr;   // the producer's handle + the consumer's operand entries, nothing else
        for (auto &o : ops) {
            if (o->done || o->kind != kind || o->in.size() != polys) continue;
            bool all = true;
            for (size_t h = 0; h < polys && all; h++) all = o->in[h].first == ph && o->in[h].second == h * w;
            if (all) return o.get();
        }
        return nullptr;
    };
    for (PendingOp *m : g_all) {
        PendingOp *r = consumer_of(m->out, OpKind::Relin, 3);
        PendingOp *d = (r && r->L == L && (!r->bgv || r->L0 == L)) ? consumer_of(r->out, OpKind::Drop, 2) : nullptr;
        const bool ok = d && d->L == L && d->bgv == r->bgv && r->rank == m->rank && d->rank == m->rank && (relin.empty() || (r->same_signature(*relin[0]) && d->same_signature(*drop[0])));
        if (!ok) {
            rest.push_back(m);
            continue;
        }
        g.push_back(m);
        relin.push_back(r);
        drop.push_back(d);
    }
    if (g.empty()) return rest;
    const PendingOp &r0 = *relin[0], &d0 = *drop[0];
    const size_t B = g.size();
    BlockRef big = alloc_block(B * d0.out_words);
    track_write(*big);
    // the operands are read where they lie (the tensor product takes their addresses): no gather of the 4 L limbs per pair
    std::vector<const u64 *> polys;
    std::vector<Src> holds;
    polys.reserve(4 * B);
    for (PendingOp *m : g)
        for (size_t h = 0; h < 4; h++) {
            holds.push_back(here(m->in[h].first, m->in[h].second, w));
            polys.push_back(holds.back().p);
        }
    const u64 *key = key_here(r0.key);
    if (r0.bgv) check(hp_dev_bgv_mult_relin_modswitch_rows(cur(), r0.logn, L, r0.mod.data(), d0.t, B, polys.data(), key, big->p));
    else check(hp_dev_ckks_mult_relin_rescale_rows(cur(), r0.logn, L, r0.L0, r0.mod.data(), B, polys.data(), key, big->p));
    for (size_t b = 0; b < B; b++) {
        DevBlock &ph = *drop[b]->out;
        ph.p = big->p + b * d0.out_words;
        ph.parent = big;
        ph.op = nullptr;
        for (PendingOp *o : {g[b], relin[b]}) {   // never materialised, and nobody can ask: see above
            o->out->op = nullptr;
            o->out->failed = true;
            o->done = true;
            release_operands(*o);
        }
        drop[b]->done = true;
        release_operands(*drop[b]);
    }
    g_stats.deferred_groups++;
    g_stats.deferred_calls += 3 * B;
    g_stats.deferred_fused += B;
    return rest;
}


// A chain of sums: add / sub calls each of which takes the previous one's result as its FIRST operand, that result held by nobody
// else (`acc = add(acc, term)` in a loop: src/circuits/linear_algebra.h:117-121, examples/ckks_example.cpp), the other operands
// ready.  The chain runs as one pass over its terms (hp_dev_poly_fold_rows: each step the lazy sum / difference of the single call, in
// the calls' order -- the same words), the intermediate sums never exist.  Returns the calls of the group that head no such chain.
std::vector<PendingOp *> run_sum_chains(const std::vector<std::unique_ptr<PendingOp>> &ops, const std::vector<PendingOp *> &g_all) {
    std::vector<PendingOp *> rest;
    if (g_all.empty() || g_all[0]->kind != OpKind::AddSub) return g_all;
    const size_t n = (size_t)1 << g_all[0]->logn, L = g_all[0]->L, w = L * n;
    auto next_of = [&](const PendingOp &t) -> PendingOp * {
        if ((size_t)t.out.use_count() != 1 + 2) return nullptr;   // the producer's handle + the consumer's two operand entries, nothing else
        for (auto &o : ops) {
            if (o->done || o->kind != OpKind::AddSub || o->in.size() != 4 || o.get() == &t) continue;
            if (o->in[0].first != t.out || o->in[0].second != 0 || o->in[1].first != t.out || o->in[1].second != w) continue;
            if (o->rank != t.rank || o->logn != t.logn || o->L != t.L || o->in_limbs != t.in_limbs || o->mod != t.mod) return nullptr;
            if (o->in[2].first->op || o->in[3].first->op) return nullptr;   // its other operand has not been computed yet
            return o.get();
        }
        return nullptr;
    };
    for (PendingOp *m : g_all) {
        std::vector<PendingOp *> chain{m};
        if (m->in_limbs == L)
            while (PendingOp *c = next_of(*chain.back())) chain.push_back(c);
        if (chain.size() < 2) {
            rest.push_back(m);
            continue;
        }
        const size_t terms = chain.size() + 1;
        std::vector<const u64 *> rows(2 * terms);
        std::vector<unsigned char> neg(terms, 0);
        std::vector<Src> holds;
        for (size_t h = 0; h < 2; h++) {
            holds.push_back(here(m->in[h].first, m->in[h].second, w));
            rows[h * terms] = holds.back().p;
            for (size_t j = 0; j < chain.size(); j++) {
                const auto &r = chain[j]->in[2 + h];
                holds.push_back(here(r.first, r.second, w));
                rows[h * terms + 1 + j] = holds.back().p;
                neg[1 + j] = chain[j]->sub ? 1 : 0;
            }
        }
        BlockRef big = alloc_block(2 * w);
        track_write(*big);
        check(hp_dev_poly_fold_rows(cur(), n, L, m->mod.data(), 2, terms, neg.data(), rows.data(), big->p));
        for (size_t j = 0; j < chain.size(); j++) {
            PendingOp *o = chain[j];
            if (j + 1 == chain.size()) {
                o->out->p = big->p;
                o->out->parent = big;
                o->out->op = nullptr;
            } else {   // never materialised, and nobody can ask: see above
                o->out->op = nullptr;
                o->out->failed = true;
            }
            o->done = true;
        }
        for (PendingOp *o : chain) release_operands(*o);
        g_stats.deferred_groups++;
        g_stats.deferred_calls += chain.size();
        g_stats.deferred_chain_sums += chain.size();
    }
    return rest;
}

} // namespace

// nothing is recorded any more: no block is waiting to be read by a recorded call
static void release_reads(const std::vector<std::unique_ptr<PendingOp>> &ops) {
    for (auto &o : ops)
        for (auto &r : o->in) {
            r.first->pending_reads = 0;
            if (r.first->parent) r.first->parent->pending_reads = 0;
        }
}

// Run everything that has been recorded: repeatedly take the oldest call that has not run (its operands are ready: whatever
// produced them was recorded earlier) and every later call with the same signature whose operands are ready too, as one batch.
void flush_all() {
    OpQueue &Q = op_queue();
    if (Q.flushing || Q.ops.empty()) return;
    Q.flushing = true;
    std::vector<std::unique_ptr<PendingOp>> ops;
    ops.swap(Q.ops);
    try {
        size_t first = 0;
        while (first < ops.size()) {
            if (ops[first]->done) { first++; continue; }
            std::vector<PendingOp *> g{ops[first].get()};
            for (size_t j = first + 1; j < ops.size(); j++)
                if (!ops[j]->done && ops[j]->same_signature(*g[0]) && ops[j]->ready()) g.push_back(ops[j].get());
            // everything the queue runs goes to lane 0: a batch fills the GPU by itself and only one lane grows a batch-sized workspace.
            // (Groups of one or two calls spread over the lanes like eager calls were measured: a dependent chain of rotations 0.163
            // against 0.126 ms per call -- the hops cost more than independent small groups could win.)
            OpScope scope({}, 0, g[0]->rank);   // (lane 0 of the rank the group was recorded for)
            g = run_fused_mults(ops, g);
            g = run_sum_chains(ops, g);
            if (!g.empty()) run_group(g);
        }
    } catch (...) {
        for (auto &o : ops)
            if (!o->done) { o->out->op = nullptr; o->out->failed = true; }   // their results throw when somebody asks for words
        release_reads(ops);
        Q.flushing = false;
        throw;
    }
    release_reads(ops);
    Q.flushing = false;
}

// record a call: returns the placeholder of its result words
BlockRef record(std::unique_ptr<PendingOp> op) {
    OpQueue &Q = op_queue();
    BlockRef ph(new DevBlock, [](DevBlock *b) { delete b; });
    ph->words = op->out_words;
    ph->rank = op->rank;
    ph->op = op.get();
    op->out = ph;
    for (auto &r : op->in) {   // (views of a batch block share its words: counted on the view and on the block that owns them)
        r.first->pending_reads++;
        if (r.first->parent) r.first->parent->pending_reads++;
    }
    Q.ops.push_back(std::move(op));
    if (Q.ops.size() >= OpQueue::MAX_PENDING) flush_all();
    return ph;
}

void set_deferred(bool on) {
#ifdef HEHUB_AMD_BIND_REFERENCE
    (void)on;
#else
    if (!on) flush_all();
    op_queue().on = on;
#endif
}

// upload a vector's host words now; both copies stay current (an operand that is read call after call -- an encoded diagonal of
// src/circuits/linear_algebra.h:111-116 -- then never crosses PCIe again, and copies of it are made on the device)
void prefetch(const RnsIntVec &v) {
#ifdef HEHUB_AMD_BIND_REFERENCE
    (void)v;   // (hehub's own objects: the device side is a cache keyed by their words, filled by the first call that reads them)
#else
    if (v.component_count() == 0) return;
    OpScope op({});
    (void)Access::in(v, v.component_count());
#endif
}

} // namespace amd

using amd::Access;
using amd::check;
using amd::OpScope;
using amd::Dst;
using amd::Src;

#ifndef HEHUB_AMD_BIND_REFERENCE
// =====================================================================================================
// rns.h: the vector itself (own mirror)
// =====================================================================================================
RnsIntVec::RnsIntVec(size_t dimension, size_t components, const std::vector<u64> &moduli) {
    size_t lg = 0;
    while (((size_t)1 << lg) < dimension) lg++;
    if (dimension == 0 || dimension != (size_t)1 << lg) throw std::invalid_argument("dimension should be a 2-power.");   // rns.cpp:17-22
    if (moduli.size() < components) throw std::invalid_argument("No matching number of moduli provided to create RnsIntVec.");
    logn_ = lg;
    count_ = components;
    q_.assign(moduli.begin(), moduli.begin() + components);
    limbs_.assign(components, ComponentData(dimension));
}
RnsIntVec::RnsIntVec(const RnsIntVec::Params &p) : RnsIntVec(p.dimension, p.component_count, p.moduli) {}
RnsIntVec::RnsIntVec(const RnsIntVec &o) { Access::copy_from(*this, o); }
RnsIntVec::RnsIntVec(RnsIntVec &&o) noexcept { Access::steal(*this, o); }
RnsIntVec &RnsIntVec::operator=(const RnsIntVec &o) {
    if (this != &o) Access::copy_from(*this, o);
    return *this;
}
RnsIntVec &RnsIntVec::operator=(RnsIntVec &&o) noexcept {
    if (this != &o) Access::steal(*this, o);
    return *this;
}
std::vector<RnsIntVec::ComponentData> &RnsIntVec::host_rw() {
    Access::host_written(*this);
    return limbs_;
}
const std::vector<RnsIntVec::ComponentData> &RnsIntVec::host_ro() const {
    Access::sync_host(*this);
    return limbs_;
}
const RnsIntVec::ComponentData &RnsIntVec::host_ro_limb(int k) const {
    Access::sync_host_limb(*this, (size_t)k);
    return limbs_[k];
}
bool RnsIntVec::operator==(const RnsIntVec &o) const {
    return logn_ == o.logn_ && count_ == o.count_ && q_ == o.q_ && host_ro() == o.host_ro();
}

void RnsIntVec::add_components(const std::vector<u64> &new_moduli, size_t adding) {
    if (new_moduli.size() < adding) throw std::invalid_argument("No matching number of moduli provided to add components.");
    host_rw();   // the new limbs are host words (zero): the host copy becomes the current one
    blk_.reset();
    off_ = 0;
    q_.insert(q_.end(), new_moduli.begin(), new_moduli.end());   // rns.cpp:41: every supplied modulus is appended, `adding` limbs are
    limbs_.insert(limbs_.end(), adding, ComponentData(dimension()));
    count_ += adding;
}

void RnsIntVec::remove_components(size_t removing) {
    if (component_count() < removing) throw std::invalid_argument("Trying to remove components more than existing.");
    q_.resize(q_.size() - removing);
    count_ -= removing;                       // both copies keep their first limbs: a device view simply gets shorter
    if (host_ok_) limbs_.resize(count_);
    stamp_ = 0;
}
#endif

namespace {

// a result polynomial of the given shape (no host words are allocated for it in the own-mirror build)
RnsPolynomial result_poly(size_t n, size_t limbs, const std::vector<u64> &moduli, PolyRepForm form) {
    RnsPolynomial p;
    Access::shape(p, n, limbs, moduli);
    p.rep_form = form;
    return p;
}

// rns.cpp:59-72 shared precondition of += and -=
size_t check_addsub(const RnsIntVec &self, const RnsIntVec &b) {
    if (self.dimension() != b.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    if (b.component_count() < self.component_count())
        throw std::invalid_argument("Operand b contains less components than self.");
    auto components = self.component_count();
    auto moduli(self.modulus_vec()), b_moduli(b.modulus_vec());
    b_moduli.resize(components);
    if (moduli != b_moduli) throw std::invalid_argument("Operands' moduli mismatch.");
    return components;
}

enum class Bin { add, sub, mul };
#ifndef HEHUB_AMD_BIND_REFERENCE
std::unique_ptr<amd::PendingOp> new_op(amd::OpKind kind, size_t logn, size_t L, const std::vector<u64> &mod,
                                       std::initializer_list<const RnsIntVec *> operands, size_t in_limbs, size_t out_words);
#endif

void dev_binary(Bin op, size_t n, size_t L, const u64 *m, size_t batch, const u64 *a, const u64 *b, u64 *out) {
    auto *ctx = amd::cur();
    if (op == Bin::add) check(hp_dev_poly_add(ctx, n, L, m, batch, a, b, out));
    if (op == Bin::sub) check(hp_dev_poly_sub(ctx, n, L, m, batch, a, b, out));
    if (op == Bin::mul) check(hp_dev_poly_mul(ctx, n, L, m, batch, a, b, out));
}

// self (op)= b on the first L limbs, in place
void run_inplace(Bin op, RnsIntVec &self, const RnsIntVec &b, size_t L) {
    const size_t n = self.dimension();
    if (L == 0 || n == 0) return;
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (op != Bin::mul && n >= 2 && amd::deferred()) {   // recorded: self becomes the placeholder of the sum (the plaintext sums of a loop run as one batch)
        OpScope scope({Access::home(self), Access::home(b)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::PolyAddSub, lg, L, self.modulus_vec(), {&self, &b}, L, L * n);
        rec->sub = op == Bin::sub;
        Access::bind_block(self, amd::record(std::move(rec)), 0);
        return;
    }
#endif
    OpScope scope({Access::home(self), Access::home(b)});
    Src sb = Access::in(b, L);
#ifndef HEHUB_AMD_BIND_REFERENCE
    u64 *p = Access::inout(self);   // the vector's own device words: hp_dev_poly_* allow d_out == d_a
    dev_binary(op, n, L, self.modulus_vec().data(), 1, p, sb.p, p);
#else
    Src sa = Access::in(self, L);
    Dst d(L * n);
    dev_binary(op, n, L, self.modulus_vec().data(), 1, sa.p, sb.p, d.p);
    Access::bind(self, d, 0, L);
#endif
}

void scalar_mul(RnsIntVec &self, const std::vector<u64> &scalars) {
    const size_t n = self.dimension(), L = self.component_count();
    if (L == 0) return;
    OpScope op({Access::home(self)});
#ifndef HEHUB_AMD_BIND_REFERENCE
    u64 *p = Access::inout(self);
    check(hp_dev_poly_scalar_mul(amd::cur(), n, L, self.modulus_vec().data(), 1, scalars.data(), p, p));
#else
    Src s = Access::in(self, L);
    Dst d(L * n);
    check(hp_dev_poly_scalar_mul(amd::cur(), n, L, self.modulus_vec().data(), 1, scalars.data(), s.p, d.p));
    Access::bind(self, d, 0, L);
#endif
}

void check_ct_wellformed(const RlweCt &ct) {   // rescaling.cpp:15-29, mod_switch.cpp:14-28
    if (ct[0].modulus_vec() != ct[1].modulus_vec())
        throw std::invalid_argument("Ill-formed ciphertext: modulus sets mismatch.");
    if (ct[0].dimension() != ct[1].dimension())
        throw std::invalid_argument("Ill-formed ciphertext: polynomial lengths mismatch.");
    if (ct[0].component_count() != ct[1].component_count())
        throw std::invalid_argument("Ill-formed ciphertext: component numbers mismatch.");
    if (ct[0].component_count() == 1) throw std::invalid_argument("Unable to drop the only one prime.");
}

// HEHUB_AMD_EXTENSIONS=1 turns on what hehub itself throws for (include/hehub_amd.h "extensions"): a key generated
// for more ciphertext moduli than the operand has, and rescale_inplace by several primes
bool extensions_on() {
    static const bool on = std::getenv("HEHUB_AMD_EXTENSIONS") && std::atoi(std::getenv("HEHUB_AMD_EXTENSIONS")) != 0;
    return on;
}

// rgsw.cpp:58-89; returns the number of ciphertext moduli the key was generated for (== pt's unless extensions are on)
size_t check_ext_prod(const RlwePt &pt, const RgswCt &rgsw, std::vector<u64> &extended_moduli) {
    if (rgsw.empty()) throw std::invalid_argument("Empty RGSW ciphertext.");
    extended_moduli = rgsw[0][0].modulus_vec();
    const auto original = pt.component_count();
    const auto extended = original + 1;
    if (extended_moduli.size() < extended) throw std::invalid_argument("Invalid component number in RGSW ciphertext.");
    const std::vector<u64> key_moduli(extended_moduli);
    extended_moduli.resize(extended);
    *extended_moduli.rbegin() = *rgsw[0][0].modulus_vec().crbegin();
    for (size_t i = 0; i < original; i++)
        if (extended_moduli[i] != pt.modulus_at((int)i)) throw std::invalid_argument("Moduli mismatch.");
    const bool higher = extensions_on() && rgsw.size() > original;
    for (auto &sample : rgsw)
        for (auto &poly : sample) {
            if (poly.dimension() != pt.dimension()) throw std::invalid_argument("Polynomial lengths mismatch.");
            if (higher ? (poly.component_count() != rgsw.size() + 1 || poly.modulus_vec() != key_moduli)
                       : (poly.component_count() != extended || poly.modulus_vec() != extended_moduli))
                throw std::invalid_argument("Inconsistent RGSW ciphertext.");
        }
    if (!higher && rgsw.size() != original) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    return rgsw.size();
}

// Device copy of a key-switching key: u64[L][2][L+1][N], one block, assembled from the key's 2L polynomials.  A key is
// 2L(L+1) limbs (55 MiB at N=32768, L=10) and the same object call after call, so assembling it every time dominates.
//   own mirror:        the last HEHUB_AMD_KEY_CACHE (default 64) keys stay resident, recognised EXACTLY: every polynomial
//                      carries a stamp that changes whenever its words can have changed (RnsIntVec, hehub.hpp)
//   binding hehub's:   with HEHUB_AMD_KEY_CACHE=<entries> keys stay resident, recognised by the address of their first
//                      limb, their shape and four sampled words of every limb (a key that is modified in place between
//                      calls without touching any sampled word would go unnoticed: that is why the cache is opt-in there);
//                      default: the key is staged per call
class DevKey {
public:
    DevKey(const RgswCt &rgsw, size_t L, size_t n) {
#ifndef HEHUB_AMD_BIND_REFERENCE
        // (64: a rotation key set stays resident -- 3.4 GiB of 288 at the C3 shape)
        static const size_t cap = std::getenv("HEHUB_AMD_KEY_CACHE") ? (size_t)std::atoi(std::getenv("HEHUB_AMD_KEY_CACHE")) : 64;
#else
        static const size_t cap = std::getenv("HEHUB_AMD_KEY_CACHE") ? (size_t)std::atoi(std::getenv("HEHUB_AMD_KEY_CACHE")) : 0;
#endif
        const size_t words = L * 2 * (L + 1) * n;
        if (cap == 0) {
            own_ = amd::alloc_block(words);
            amd::track_write(*own_);
            assemble(own_->p, rgsw, L, n);
            return;
        }
        std::vector<u64> sig{(u64)L, (u64)n, (u64)amd::cur_rank()};   // (a key is assembled once per device rank that uses it)
        for (size_t j = 0; j < L; j++)
            for (size_t h = 0; h < 2; h++) {
#ifndef HEHUB_AMD_BIND_REFERENCE
                sig.push_back(Access::stamp(rgsw[j][h]));
#else
                sig.push_back((u64)(uintptr_t)rgsw[j][h][0].data());
                for (size_t k = 0; k <= L; k++) {
                    const u64 *w = rgsw[j][h][(int)k].data();
                    sig.insert(sig.end(), {w[0], w[n / 3], w[(2 * n) / 3], w[n - 1]});
                }
#endif
            }
        typedef std::vector<std::pair<std::vector<u64>, amd::BlockRef>> Cache;   // most recently used last
        static std::mutex &mu = *new std::mutex;   // (never destroyed, like the pool)
        static Cache &cache = *new Cache;
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < cache.size(); i++)
            if (cache[i].first == sig) {
                auto hit = cache[i];
                cache.erase(cache.begin() + i);
                cache.push_back(hit);
                own_ = hit.second;
                amd::track_read(*own_);
                return;
            }
        own_ = amd::alloc_block(words);
        amd::track_write(*own_);
        assemble(own_->p, rgsw, L, n);
        if (cache.size() >= cap) cache.erase(cache.begin());
        cache.emplace_back(std::move(sig), own_);
    }
    const u64 *p() const { return own_->p; }
    const amd::BlockRef &block() const { return own_; }

private:
    static void assemble(u64 *dst, const RgswCt &rgsw, size_t L, size_t n) {
        for (size_t j = 0; j < L; j++)
            for (size_t h = 0; h < 2; h++) {
                u64 *row = dst + ((j * 2 + h) * (L + 1)) * n;
#ifndef HEHUB_AMD_BIND_REFERENCE
                Src s = Access::in(rgsw[j][h], L + 1);   // (a key polynomial that lives on the device is copied there)
                check(hp_dev_copy(amd::cur(), (L + 1) * n, s.p, row));
                Access::drop_device_copy(rgsw[j][h]);     // the assembled block is the key's device form: no second 55 MiB
#else
                amd::poly_copy_h2d(row, rgsw[j][h], L + 1, n);
#endif
            }
    }
    amd::BlockRef own_;
};

#ifndef HEHUB_AMD_BIND_REFERENCE
// deferred mode: a recorded call over the given operand polynomials (host words are uploaded now, placeholders stay placeholders)
std::unique_ptr<amd::PendingOp> new_op(amd::OpKind kind, size_t logn, size_t L, const std::vector<u64> &mod,
                                       std::initializer_list<const RnsIntVec *> operands, size_t in_limbs, size_t out_words) {
    std::unique_ptr<amd::PendingOp> op(new amd::PendingOp);
    op->kind = kind; op->logn = logn; op->L = L; op->L0 = L; op->mod = mod; op->in_limbs = in_limbs; op->out_words = out_words;
    op->rank = amd::rank_of(amd::lane_set().cur);   // (the recording scope chose the rank from the operands: RecordScope)
    for (const RnsIntVec *v : operands) op->in.push_back(Access::ref(*v));
    return op;
}
// the polynomials of a recorded call's result: views [h * limbs * N, ..) of its placeholder
template <class Polys> void bind_placeholder(Polys &polys, size_t count, const amd::BlockRef &ph, size_t limbs, size_t n) {
    for (size_t h = 0; h < count; h++) Access::bind_block(polys[h], ph, h * limbs * n);
}
#endif

// the two polynomials of a result ciphertext as views of the block an engine call filled: [2][L][N]
RlweCt make_ct(size_t n, size_t L, const std::vector<u64> &moduli, const Dst &d) {
    RlweCt ct{result_poly(n, L, moduli, PolyRepForm::value), result_poly(n, L, moduli, PolyRepForm::value)};
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 2; h++) Access::bind_enqueue(ct[h], d, (size_t)h * L * n, L);
    amd::limb_copies_wait();
    for (int h = 0; h < 2; h++) Access::bind_finish(ct[h], d, (size_t)h * L * n, L);
#else
    for (int h = 0; h < 2; h++) Access::bind(ct[h], d, (size_t)h * L * n, L);
#endif
    return ct;
}

// shared body of ckks::relinearize / bgv::relinearize
RlweCt relinearize_common(const std::array<RnsPolynomial, 3> &quad, const RlweKsk &key, bool bgv) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(quad[2], key, mext);
    const size_t n = quad[2].dimension(), L = quad[2].component_count();
    if (bgv && L0 != L) throw std::invalid_argument("Inconsistent RGSW ciphertext.");   // no higher-level keys for the BGV quirk path
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(quad[0]), Access::home(quad[1]), Access::home(quad[2])}, 0);
        DevKey dk(key, L0, n);
        auto rec = new_op(amd::OpKind::Relin, quad[2].log_dimension(), L, mext, {&quad[0], &quad[1], &quad[2]}, L, 2 * L * n);
        rec->L0 = L0; rec->bgv = bgv; rec->key = dk.block();
        std::vector<u64> q(mext.begin(), mext.begin() + L);
        RlweCt ct{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)};
        bind_placeholder(ct, 2, amd::record(std::move(rec)), L, n);
        return ct;
    }
#endif
    OpScope op({Access::home(quad[0]), Access::home(quad[1]), Access::home(quad[2])});
    DevKey dk(key, L0, n);
    Src dq = amd::gather({&quad[0], &quad[1], &quad[2]}, L);
    Dst dout(2 * L * n);
    const size_t logn = quad[2].log_dimension();
    if (bgv) check(hp_dev_bgv_relinearize(amd::cur(), logn, L, mext.data(), 1 /* bgv.h:32 */, 1, dq.p, dk.p(), dout.p));
    else check(hp_dev_ckks_relinearize_at(amd::cur(), logn, L, L0, mext.data(), 1, dq.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    return make_ct(n, L, q, dout);
}

template <class Quad, class Ct> Quad mult_low_level_common(const Ct &ct1, const Ct &ct2) {
    for (int h = 0; h < 2; h++) {
        if (ct1[h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
        if (ct2[h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    }
    if (ct1[0].dimension() != ct2[0].dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t n = ct1[0].dimension();
    const size_t L = std::min(ct1[0].component_count(), ct2[0].component_count());
    std::vector<u64> m1(ct1[0].modulus_vec()), m2(ct2[0].modulus_vec());
    m1.resize(L); m2.resize(L);
    if (m1 != m2) throw std::invalid_argument("Operands' moduli mismatch.");
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct1[0]), Access::home(ct1[1]), Access::home(ct2[0]), Access::home(ct2[1])}, 0);
        auto rec = new_op(amd::OpKind::MultLow, ct1[0].log_dimension(), L, m1, {&ct1[0], &ct1[1], &ct2[0], &ct2[1]}, L, 3 * L * n);
        Quad quad;
        for (int h = 0; h < 3; h++) quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        bind_placeholder(quad, 3, amd::record(std::move(rec)), L, n);
        return quad;
    }
#endif
    // (a ciphertext with more limbs than L does not lie as [2][L][N]: gather() then copies the first L limbs of each half)
    OpScope op({Access::home(ct1[0]), Access::home(ct1[1]), Access::home(ct2[0]), Access::home(ct2[1])});
    Src d1 = amd::gather({&ct1[0], &ct1[1]}, L);
    Src d2 = amd::gather({&ct2[0], &ct2[1]}, L);
    Dst dq(3 * L * n);
    check(hp_dev_mult_low_level(amd::cur(), ct1[0].log_dimension(), L, m1.data(), 1, d1.p, d2.p, dq.p));
    Quad quad;
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 3; h++) {
        quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        Access::bind_enqueue(quad[h], dq, (size_t)h * L * n, L);
    }
    amd::limb_copies_wait();
    for (int h = 0; h < 3; h++) Access::bind_finish(quad[h], dq, (size_t)h * L * n, L);
#else
    for (int h = 0; h < 3; h++) {
        quad[h] = result_poly(n, L, m1, PolyRepForm::value);
        Access::bind(quad[h], dq, (size_t)h * L * n, L);
    }
#endif
    return quad;
}

void drop_last_prime(RlweCt &ct, bool bgv, u64 t) {
    check_ct_wellformed(ct);
    const size_t n = ct[0].dimension(), L = ct[0].component_count(), logn = ct[0].log_dimension();
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct[0]), Access::home(ct[1])}, 0);
        auto rec = new_op(amd::OpKind::Drop, logn, L, ct[0].modulus_vec(), {&ct[0], &ct[1]}, L, 2 * (L - 1) * n);
        rec->bgv = bgv; rec->t = t;
        const amd::BlockRef ph = amd::record(std::move(rec));
        for (int h = 0; h < 2; h++) {
            ct[h].remove_components();
            Access::bind_block(ct[h], ph, (size_t)h * (L - 1) * n);
        }
        return;
    }
#endif
    OpScope op({Access::home(ct[0]), Access::home(ct[1])});
    Src din = amd::gather({&ct[0], &ct[1]}, L);
    Dst dout(2 * (L - 1) * n);
    const std::vector<u64> m(ct[0].modulus_vec());
    if (bgv) check(hp_dev_bgv_mod_switch(amd::cur(), logn, L, m.data(), t, 1, din.p, dout.p));
    else check(hp_dev_ckks_rescale(amd::cur(), logn, L, m.data(), 1, din.p, dout.p));
#ifdef HEHUB_AMD_BIND_REFERENCE
    // remove_components hands the last limb's block back to hehub's pool, whose free list writes its link into the block's first
    // word (allocator.h:67-71): the (asynchronous) upload of that limb must have happened by then
    amd::limb_copies_wait();
#endif
#ifdef HEHUB_AMD_BIND_REFERENCE
    for (int h = 0; h < 2; h++) {
        ct[h].remove_components();
        Access::bind_enqueue(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
    }
    amd::limb_copies_wait();
    for (int h = 0; h < 2; h++) Access::bind_finish(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
#else
    for (int h = 0; h < 2; h++) {
        ct[h].remove_components();
        Access::bind(ct[h], dout, (size_t)h * (L - 1) * n, L - 1);
    }
#endif
}

} // namespace

// =====================================================================================================
// rns.h / rns.cpp: operators
// =====================================================================================================
const RnsIntVec &operator+=(RnsIntVec &self, const RnsIntVec &b) {
    run_inplace(Bin::add, self, b, check_addsub(self, b));
    return self;
}

const RnsIntVec &operator-=(RnsIntVec &self, const RnsIntVec &b) {
    run_inplace(Bin::sub, self, b, check_addsub(self, b));
    return self;
}

RnsIntVec operator*(const RnsIntVec &a, const RnsIntVec &b) {
    if (a.dimension() != b.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    auto components = std::min(a.component_count(), b.component_count());
    auto moduli(a.modulus_vec()), b_moduli(b.modulus_vec());
    moduli.resize(components);
    b_moduli.resize(components);
    if (moduli != b_moduli) throw std::invalid_argument("Operands' moduli mismatch.");
    RnsIntVec result;
    Access::shape(result, a.dimension(), components, moduli);
    const size_t n = a.dimension();
    if (components == 0 || n == 0) return result;
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred() && n >= 2) {
        OpScope scope({Access::home(a), Access::home(b)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::PolyMul, lg, components, moduli, {&a, &b}, components, components * n);
        Access::bind_block(result, amd::record(std::move(rec)), 0);
        return result;
    }
#endif
    OpScope op({Access::home(a), Access::home(b)});
    Src sa = Access::in(a, components), sb = Access::in(b, components);
    Dst d(components * n);
    dev_binary(Bin::mul, n, components, moduli.data(), 1, sa.p, sb.p, d.p);
    Access::bind(result, d, 0, components);
    return result;
}

const RnsIntVec &operator*=(RnsIntVec &self, const u64 small_scalar) {
    scalar_mul(self, std::vector<u64>(self.component_count(), small_scalar));
    return self;
}

const RnsIntVec &operator*=(RnsIntVec &self, const std::vector<u64> &rns_scalar) {
    if (rns_scalar.size() != self.component_count()) throw std::invalid_argument("Numbers of RNS component mismatch.");
    scalar_mul(self, rns_scalar);
    return self;
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // inline in the reference's rns.h:207-293
const RnsPolynomial &operator+=(RnsPolynomial &self, const RnsPolynomial &b) {
    if (self.rep_form != b.rep_form) throw std::invalid_argument("Operands are in different representation form.");
    (RnsIntVec &)self += (const RnsIntVec &)b;
    return self;
}

const RnsPolynomial &operator-=(RnsPolynomial &self, const RnsPolynomial &b) {
    if (self.rep_form != b.rep_form) throw std::invalid_argument("Operands are in different representation form.");
    (RnsIntVec &)self -= (const RnsIntVec &)b;
    return self;
}

RnsPolynomial operator*(const RnsPolynomial &a, const RnsPolynomial &b) {
    if (a.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
    if (b.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    RnsPolynomial result = (const RnsIntVec &)a * (const RnsIntVec &)b;
    result.rep_form = PolyRepForm::value;
    return result;
}

const RnsPolynomial &operator*=(RnsPolynomial &self, const u64 s) {
    (RnsIntVec &)self *= s;
    return self;
}

const RnsPolynomial &operator*=(RnsPolynomial &self, const std::vector<u64> &s) {
    (RnsIntVec &)self *= s;
    return self;
}
#endif

// =====================================================================================================
// mod_arith.h
// =====================================================================================================
void batched_barrett_lazy(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett_lazy(amd::cur(), q, n, v)); }
#ifndef HEHUB_AMD_BIND_REFERENCE   // inline in the reference's mod_arith.h:18-25,58-63
void batched_barrett(const u64 q, const size_t n, u64 v[]) { check(hp_batched_barrett(amd::cur(), q, n, v)); }
void batched_reduce_strict(const u64 q, const size_t n, u64 v[]) { check(hp_batched_reduce_strict(amd::cur(), q, n, v)); }
#endif
void batched_mul_mod_hybrid_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_hybrid_lazy(amd::cur(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_hybrid(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_hybrid_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_mul_mod_barrett_lazy(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    check(hp_batched_mul_mod_barrett_lazy(amd::cur(), q, n, a, b, out));
}
#ifndef HEHUB_AMD_BIND_REFERENCE
void batched_mul_mod_barrett(const u64 q, const size_t n, const u64 a[], const u64 b[], u64 out[]) {
    batched_mul_mod_barrett_lazy(q, n, a, b, out);
    batched_reduce_strict(q, n, out);
}
#endif
void batched_montgomery_128_lazy(const u64 q, const size_t len, const u128 in[], u64 out[]) {
    check(hp_batched_montgomery_128_lazy(amd::cur(), q, len, reinterpret_cast<const u64 *>(in), out));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // mod_arith.h:65-72 (inline) and mod_arith.cpp:136-149 stay the reference's
void reduce_strict(RnsPolynomial &p) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L == 0) return;
    OpScope op({Access::home(p)});
    check(hp_dev_poly_reduce_strict(amd::cur(), n, L, p.modulus_vec().data(), 1, Access::inout(p)));
}

// host-side scalar, as in the reference (mod_arith.cpp:136-149): Bezout coefficient lifted to [0, prime)
u64 inverse_mod_prime(const u64 elem, const u64 prime) {
    __int128 r0 = prime, r1 = elem, y0 = 0, y1 = 1;
    while (r1 != 0) {
        __int128 quo = r0 / r1, r2 = r0 - quo * r1, y2 = y0 - quo * y1;
        r0 = r1; r1 = r2; y0 = y1; y1 = y2;
    }
    if (y0 < 0) y0 += prime;
    return (u64)y0;
}
#endif

// =====================================================================================================
// ntt.h
// =====================================================================================================
void ntt_negacyclic_inplace_lazy(const size_t logn, const u64 q, u64 c[]) {
    check(hp_ntt_negacyclic_inplace_lazy(amd::cur(), logn, q, c));
}
void intt_negacyclic_inplace_lazy(const size_t logn, const u64 q, u64 v[]) {
    check(hp_intt_negacyclic_inplace_lazy(amd::cur(), logn, q, v));
}

#ifndef HEHUB_AMD_BIND_REFERENCE   // ntt.h:41-92 (inline per-limb loops) stay the reference's
static void poly_transform(RnsPolynomial &p, bool inverse, bool strict) {
    const size_t n = p.dimension(), L = p.component_count();
    if (L && n >= 2 && amd::deferred()) {   // recorded: the transforms of a loop's plaintexts run as one batch (conj = inverse, sub = strict)
        OpScope scope({Access::home(p)}, 0);
        auto rec = new_op(amd::OpKind::Transform, p.log_dimension(), L, p.modulus_vec(), {&p}, L, L * n);
        rec->conj = inverse; rec->sub = strict;
        Access::bind_block(p, amd::record(std::move(rec)), 0);
    } else if (L) {
        OpScope op({Access::home(p)});
        u64 *d = Access::inout(p);
        if (inverse) check(hp_dev_intt(amd::cur(), p.log_dimension(), L, p.modulus_vec().data(), 1, d, strict ? 1 : 0));
        else check(hp_dev_ntt(amd::cur(), p.log_dimension(), L, p.modulus_vec().data(), 1, d));
    }
    p.rep_form = inverse ? PolyRepForm::coeff : PolyRepForm::value;
}
void ntt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, false, false); }
void intt_negacyclic_inplace_lazy(RnsPolynomial &p) { poly_transform(p, true, false); }
void intt_negacyclic_inplace(RnsPolynomial &p) { poly_transform(p, true, true); }
#endif

void cache_ntt_factors_strict(const u64 logn, const std::vector<u64> &moduli) {
    check(hp_cache_ntt_factors_strict(amd::cur(), logn, moduli.data(), moduli.size()));
}

// =====================================================================================================
// permutation.h
// =====================================================================================================
static RnsPolynomial gather(const RnsPolynomial &p, bool is_cycle, size_t step) {
    if (p.rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    const size_t n = p.dimension(), L = p.component_count();
    RnsPolynomial out = result_poly(n, L, p.modulus_vec(), PolyRepForm::value);
    if (L == 0) return out;
    OpScope op({Access::home(p)});
    Src din = Access::in(p, L);
    Dst dout(L * n);
    if (is_cycle) check(hp_dev_poly_cycle(amd::cur(), p.log_dimension(), L, 1, step, din.p, dout.p));
    else check(hp_dev_poly_involution(amd::cur(), p.log_dimension(), L, 1, din.p, dout.p));
    Access::bind(out, dout, 0, L);
    return out;
}
RnsPolynomial cycle(const RnsPolynomial &p, const size_t step) { return gather(p, true, step); }
RnsPolynomial involution(const RnsPolynomial &p) { return gather(p, false, 0); }

// =====================================================================================================
// rlwe.h / rgsw.h
// =====================================================================================================
#ifndef HEHUB_AMD_BIND_REFERENCE   // rlwe.cpp:83-101: thin compositions of the operators above
// both halves in ONE launch when the two ciphertexts have the same shape (the result's halves then lie side by side, ready
// for the next scheme-level call); the checks are operator+= 's (rns.h:207-218, rns.cpp:59-72), half by half, in its order
static RlweCt addsub(const RlweCt &a, const RlweCt &b, bool sub) {
    size_t L[2];
    for (int h = 0; h < 2; h++) {
        if (a[h].rep_form != b[h].rep_form) throw std::invalid_argument("Operands are in different representation form.");
        L[h] = check_addsub(a[h], b[h]);
    }
    const size_t n = a[0].dimension();
    const bool same = L[0] == L[1] && L[0] > 0 && a[1].dimension() == n && a[0].modulus_vec() == a[1].modulus_vec() &&
                      b[0].component_count() == L[0] && b[1].component_count() == L[0];
    if (same && amd::deferred()) {
        OpScope scope({Access::home(a[0]), Access::home(a[1]), Access::home(b[0]), Access::home(b[1])}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        auto rec = new_op(amd::OpKind::AddSub, lg, L[0], a[0].modulus_vec(), {&a[0], &a[1], &b[0], &b[1]}, L[0], 2 * L[0] * n);
        rec->sub = sub;
        RlweCt r{result_poly(n, L[0], a[0].modulus_vec(), a[0].rep_form), result_poly(n, L[0], a[1].modulus_vec(), a[1].rep_form)};
        bind_placeholder(r, 2, amd::record(std::move(rec)), L[0], n);
        return r;
    }
    OpScope op({Access::home(a[0]), Access::home(a[1]), Access::home(b[0]), Access::home(b[1])});
    if (!same) return sub ? RlweCt{a[0] - b[0], a[1] - b[1]} : RlweCt{a[0] + b[0], a[1] + b[1]};
    Src sa = amd::gather({&a[0], &a[1]}, L[0]), sb = amd::gather({&b[0], &b[1]}, L[0]);
    Dst d(2 * L[0] * n);
    dev_binary(sub ? Bin::sub : Bin::add, n, L[0], a[0].modulus_vec().data(), 2, sa.p, sb.p, d.p);
    RlweCt r{result_poly(n, L[0], a[0].modulus_vec(), a[0].rep_form), result_poly(n, L[0], a[1].modulus_vec(), a[1].rep_form)};
    for (int h = 0; h < 2; h++) Access::bind(r[h], d, (size_t)h * L[0] * n, L[0]);
    return r;
}
RlweCt add(const RlweCt &a, const RlweCt &b) { return addsub(a, b, false); }
RlweCt sub(const RlweCt &a, const RlweCt &b) { return addsub(a, b, true); }
RlweCt add_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] + pt, ct[1]}; }
RlweCt sub_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] - pt, ct[1]}; }
RlweCt mult_plain_core(const RlweCt &ct, const RlwePt &pt) { return RlweCt{ct[0] * pt, ct[1] * pt}; }
#endif

RlweCt ext_prod_montgomery(const RlwePt &pt, const RgswCt &rgsw) {
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(pt, rgsw, mext);
    const size_t n = pt.dimension(), L = pt.component_count();
    OpScope op({Access::home(pt)});
    DevKey dk(rgsw, L0, n);
    Src dp = Access::in(pt, L);
    Dst dout(2 * (L + 1) * n);
    check(hp_dev_ext_prod_montgomery_at(amd::cur(), pt.log_dimension(), L, L0, mext.data(), 1, dp.p, dk.p(), dout.p));
    return make_ct(n, L + 1, mext, dout);
}

// rlwe.h decrypt_core (rlwe.cpp:74-81): `c0 + c1 * sk`, INTT, reduce_strict as ONE device call instead of 3L host
// round trips; the argument checks are the ones the reference's operator* / operator+ perform, in their order.
RlwePt decrypt_core(const RlweCt &ct, const RlweSk &sk) {
    const RnsPolynomial &c0 = ct[0], &c1 = ct[1];
    if (c1.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
    if (sk.rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
    if (c1.dimension() != sk.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t Lp = std::min(c1.component_count(), sk.component_count());
    std::vector<u64> m1(c1.modulus_vec()), ms(sk.modulus_vec());
    m1.resize(Lp); ms.resize(Lp);
    if (m1 != ms) throw std::invalid_argument("Operands' moduli mismatch.");
    if (c0.rep_form != PolyRepForm::value) throw std::invalid_argument("Operands are in different representation form.");
    if (c0.dimension() != c1.dimension()) throw std::invalid_argument("Operands' poly len mismatch.");
    const size_t L = c0.component_count(), n = c0.dimension();
    if (Lp < L) throw std::invalid_argument("Operand b contains less components than self.");
    m1.resize(L);
    if (c0.modulus_vec() != m1) throw std::invalid_argument("Operands' moduli mismatch.");
    RnsPolynomial pt = result_poly(n, L, m1, PolyRepForm::coeff);
    if (L == 0) return pt;
    OpScope op({Access::home(c0), Access::home(c1)});
    Src dct = amd::gather({&c0, &c1}, L), dsk = Access::in(sk, L);
    Dst dpt(L * n);
    check(hp_dev_rlwe_decrypt_core(amd::cur(), c0.log_dimension(), L, m1.data(), 1, dct.p, dsk.p, dpt.p));
    Access::bind(pt, dpt, 0, L);
    return pt;
}

#ifndef HEHUB_AMD_BIND_REFERENCE
// rns_transform.cpp:106-127 on the device: one -> many (:11-37) and many -> one (:39-104, both branches)
RnsPolynomial rns_base_transform(RnsPolynomial in, const std::vector<u64> &new_moduli) {
    if (in.rep_form == PolyRepForm::value)
        throw std::logic_error("Trying to perform RNS base transformation on NTT values.");
    const size_t n = in.dimension(), L = in.component_count();
    // recorded in deferred mode: the plaintext lifts of a loop of bgv::add_plain / sub_plain / mult_plain (bgv/arith.cpp:17-57) run as
    // one batch.  (one -> many only: many -> one is decrypt's, whose caller looks at the words next, and its preconditions -- odd,
    // pairwise coprime moduli -- are reported by the call itself)
    if (amd::deferred() && n >= 2 && L == 1 && !new_moduli.empty() && in.modulus_at(0) >= 2) {
        OpScope scope({Access::home(in)}, 0);
        size_t lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        RnsPolynomial out = result_poly(n, new_moduli.size(), new_moduli, PolyRepForm::coeff);
        auto rec = new_op(amd::OpKind::BaseConv, lg, new_moduli.size(), new_moduli, {&in}, 1, new_moduli.size() * n);
        rec->t = in.modulus_at(0);
        Access::bind_block(out, amd::record(std::move(rec)), 0);
        return out;
    }
    OpScope op({Access::home(in)});
    if (L == 1) {
        RnsPolynomial out = result_poly(n, new_moduli.size(), new_moduli, PolyRepForm::coeff);
        if (new_moduli.empty()) return out;
        Src din = Access::in(in, 1);
        Dst dout(new_moduli.size() * n);
        check(hp_dev_rns_base_from_single(amd::cur(), n, in.modulus_at(0), new_moduli.size(), new_moduli.data(), 1, din.p, dout.p));
        Access::bind(out, dout, 0, new_moduli.size());
        return out;
    }
    if (new_moduli.size() == 1) {   // both branches of rns_transform.cpp:39-104 on the device
        RnsPolynomial out = result_poly(n, 1, new_moduli, PolyRepForm::coeff);
        Src din = Access::in(in, L);
        Dst dout(n);
        check(hp_dev_rns_base_to_single(amd::cur(), n, L, in.modulus_vec().data(), new_moduli[0], 1, din.p, dout.p));
        Access::bind(out, dout, 0, 1);
        return out;
    }
    throw "under development";   // rns_transform.cpp:123
}
#endif

// =====================================================================================================
// ckks.h
// =====================================================================================================
namespace ckks {

#ifndef HEHUB_AMD_BIND_REFERENCE   // ckks/arith.cpp:7-53 stay the reference's
static void check_scaling_factor(double a, double b) {   // ckks/arith.cpp:7-13
    if (std::abs(a - b) > std::pow(2.0, -50)) throw std::invalid_argument("The scaling factors mismatch");
}

CkksCt add(const CkksCt &a, const CkksCt &b) {
    check_scaling_factor(a.scaling_factor, b.scaling_factor);
    CkksCt r = ::hehub::add((const RlweCt &)a, (const RlweCt &)b);
    r.scaling_factor = a.scaling_factor;
    return r;
}

CkksCt sub(const CkksCt &a, const CkksCt &b) {
    check_scaling_factor(a.scaling_factor, b.scaling_factor);
    CkksCt r = ::hehub::sub((const RlweCt &)a, (const RlweCt &)b);
    r.scaling_factor = a.scaling_factor;
    return r;
}

CkksCt add_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:22-29
    check_scaling_factor(ct.scaling_factor, pt.scaling_factor);
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = add_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt sub_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:38-45
    check_scaling_factor(ct.scaling_factor, pt.scaling_factor);
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = sub_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt mult_plain(const CkksCt &ct, const CkksPt &pt) {   // ckks/arith.cpp:47-53
    RnsPolynomial pt_ntt(pt);
    ntt_negacyclic_inplace_lazy(pt_ntt);
    CkksCt r = mult_plain_core(ct, pt_ntt);
    r.scaling_factor = ct.scaling_factor * pt.scaling_factor;
    return r;
}
#endif

CkksQuadraticCt mult_low_level(const CkksCt &a, const CkksCt &b) {
    auto q = mult_low_level_common<CkksQuadraticCt>(a, b);
    q.scaling_factor = a.scaling_factor * b.scaling_factor;
    return q;
}

CkksCt relinearize(const CkksQuadraticCt &ct, const RlweKsk &key) {
    CkksCt r = relinearize_common(ct, key, false);
    r.scaling_factor = ct.scaling_factor;   // ckks/arith.cpp:68
    return r;
}

void rescale_inplace(CkksCt &ct, size_t dropping_primes) {   // rescaling.cpp:80-90
    if (dropping_primes == 1) {
        check_ct_wellformed(ct);
        const u64 q_last = *ct[0].modulus_vec().crbegin();
        drop_last_prime(ct, false, 0);
        ct.scaling_factor /= q_last;
    } else if (dropping_primes >= 2) {
        if (!extensions_on()) throw "under development";
        for (size_t d = 0; d < dropping_primes; d++) rescale_inplace(ct, 1);   // successive exact one-prime drops
    } else {
        throw std::invalid_argument("The number of primes to be dropped is not positive.");
    }
}

// ckks/arith.cpp:75-93: automorphism, key switch, drop of the special prime and the add of moved[0] run as one
// device call; the argument checks below are the ones the reference's composition performs, in its order.
static CkksCt key_switched(const CkksCt &ct, const RlweKsk &key, bool conj, size_t step) {
    for (int h = 0; h < 2; h++)
        if (ct[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(ct[1], key, mext);
    const size_t n = ct[1].dimension(), L = ct[1].component_count(), logn = ct[1].log_dimension();
    if (ct[0].dimension() != n || ct[0].component_count() != L) throw std::invalid_argument("Ill-formed ciphertext.");
#ifndef HEHUB_AMD_BIND_REFERENCE
    if (amd::deferred()) {
        OpScope scope({Access::home(ct[0]), Access::home(ct[1])}, 0);
        DevKey dk(key, L0, n);
        auto rec = new_op(amd::OpKind::KeySwitch, logn, L, mext, {&ct[0], &ct[1]}, L, 2 * L * n);
        rec->L0 = L0; rec->conj = conj; rec->step = conj ? 0 : step; rec->key = dk.block();
        std::vector<u64> q(mext.begin(), mext.begin() + L);
        CkksCt r(RlweCt{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)});
        bind_placeholder(r, 2, amd::record(std::move(rec)), L, n);
        r.scaling_factor = ct.scaling_factor;
        return r;
    }
#endif
    OpScope op({Access::home(ct[0]), Access::home(ct[1])});
    DevKey dk(key, L0, n);
    Src dct = amd::gather({&ct[0], &ct[1]}, L);
    Dst dout(2 * L * n);
    if (conj) check(hp_dev_ckks_conjugate_at(amd::cur(), logn, L, L0, mext.data(), 1, dct.p, dk.p(), dout.p));
    else check(hp_dev_ckks_rotate_at(amd::cur(), logn, L, L0, mext.data(), 1, step, dct.p, dk.p(), dout.p));
    std::vector<u64> q(mext.begin(), mext.begin() + L);
    CkksCt r = make_ct(n, L, q, dout);
    r.scaling_factor = ct.scaling_factor;
    return r;
}

CkksCt conjugate(const CkksCt &ct, const RlweKsk &conj_key) { return key_switched(ct, conj_key, true, 0); }

CkksCt rotate(const CkksCt &ct, const RlweKsk &rot_key, const size_t step) { return key_switched(ct, rot_key, false, step); }

} // namespace ckks

// =====================================================================================================
// bgv.h
// =====================================================================================================
namespace bgv {

#ifndef HEHUB_AMD_BIND_REFERENCE   // bgv/arith.cpp:8-57 stay the reference's
BgvCt add(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    BgvCt r = ::hehub::add((const RlweCt &)a, (const RlweCt &)b);
    r.plain_modulus = a.plain_modulus;
    return r;
}

BgvCt sub(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    BgvCt r = ::hehub::sub((const RlweCt &)a, (const RlweCt &)b);
    r.plain_modulus = a.plain_modulus;
    return r;
}

// bgv/arith.cpp:17-57: the plaintext (one component modulo t) is lifted into the ciphertext's moduli, transformed
// and combined with the RLWE core operation
static RnsPolynomial lift_plain(const BgvCt &ct, const BgvPt &pt) {
    if (pt.component_count() != 1 || pt.modulus_at(0) != ct.plain_modulus) throw std::invalid_argument("plain moduli mismatch.");
    auto lifted = rns_base_transform(pt, ct[0].modulus_vec());
    ntt_negacyclic_inplace_lazy(lifted);
    return lifted;
}
BgvCt add_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::add_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
BgvCt sub_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::sub_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
BgvCt mult_plain(const BgvCt &ct, const BgvPt &pt) {
    BgvCt r = ::hehub::mult_plain_core(ct, lift_plain(ct, pt));
    r.plain_modulus = ct.plain_modulus;
    return r;
}
#endif

BgvQuadraticCt mult_low_level(const BgvCt &a, const BgvCt &b) {
    if (a.plain_modulus != b.plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");
    auto q = mult_low_level_common<BgvQuadraticCt>(a, b);
    q.plain_modulus = a.plain_modulus;
    return q;
}

BgvCt relinearize(const BgvQuadraticCt &ct, const RlweKsk &key) {
    BgvCt r = relinearize_common(ct, key, true);
    r.plain_modulus = ct.plain_modulus;
    return r;
}

void mod_switch_inplace(BgvCt &ct, size_t dropping_primes) {   // mod_switch.cpp:80-90
    if (dropping_primes == 1) {
        drop_last_prime(ct, true, ct.plain_modulus);
    } else if (dropping_primes >= 2) {
        throw "under development";
    } else {
        throw std::invalid_argument("The number of primes to be dropped is not positive.");
    }
}

} // namespace bgv


// =====================================================================================================
// hehub_amd_ext.hpp: batched forms of the scheme-level calls
// =====================================================================================================
namespace amd {

namespace {

// the shape all members of a batch share (both halves of every ciphertext: dimension, limbs, moduli), or false
template <class Ct> bool uniform_shape(const std::vector<Ct> &cts, size_t &n, size_t &L, std::vector<u64> &q) {
    if (cts.empty()) return false;
    n = cts[0][0].dimension();
    L = cts[0][0].component_count();
    q = cts[0][0].modulus_vec();
    q.resize(L);
    if (L == 0 || n < 2) return false;
    for (const Ct &ct : cts)
        for (int h = 0; h < 2; h++) {
            if (ct[h].dimension() != n || ct[h].component_count() != L) return false;
            std::vector<u64> m(ct[h].modulus_vec());
            m.resize(L);
            if (m != q) return false;
        }
    return true;
}
template <class Ct> std::vector<const RnsIntVec *> halves(const std::vector<Ct> &cts) {
    std::vector<const RnsIntVec *> v;
    v.reserve(2 * cts.size());
    for (const Ct &ct : cts) { v.push_back(&ct[0]); v.push_back(&ct[1]); }
    return v;
}
// B result ciphertexts of the given shape, no words yet
template <class Ct> std::vector<Ct> result_shells(size_t B, size_t n, size_t L, const std::vector<u64> &q) {
    std::vector<Ct> out;
    out.reserve(B);
    for (size_t i = 0; i < B; i++)
        out.emplace_back(RlweCt{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)});
    return out;
}
// ... elements [lo, hi) become views of u64[hi - lo][2][L][N], the block one rank's engine call filled
template <class Ct> void bind_slice(std::vector<Ct> &out, size_t lo, size_t hi, const Dst &d, size_t L) {
    std::vector<RnsIntVec *> polys;
    for (size_t i = lo; i < hi; i++) { polys.push_back(&out[i][0]); polys.push_back(&out[i][1]); }
    Access::bind_many(polys, d, L);
}
template <class Ct> std::vector<const RnsIntVec *> halves(const std::vector<Ct> &cts, size_t lo, size_t hi) {
    std::vector<const RnsIntVec *> v;
    v.reserve(2 * (hi - lo));
    for (size_t i = lo; i < hi; i++) { v.push_back(&cts[i][0]); v.push_back(&cts[i][1]); }
    return v;
}
// A batch is cut into contiguous slices, one per device rank (batch / ranks each; with one rank: the whole batch), and
// f(lo, hi) runs for each slice inside a scope on lane 0 of its rank: a batch fills a GPU by itself, and only one lane per rank
// grows a batch-sized workspace.  The slices' engine calls are enqueued one after the other and run side by side on their devices.
template <class F> void for_slices(size_t B, F &&f) {
    (void)engine();
    const int nd = lane_set().ndev;
    for (int r = 0; r < nd; r++) {
        const size_t lo = B * (size_t)r / (size_t)nd, hi = B * (size_t)(r + 1) / (size_t)nd;
        if (lo == hi) continue;
        OpScope op({}, 0, r);
        f(lo, hi);
    }
}
void same_size(size_t a, size_t b) {
    if (a != b) throw std::invalid_argument("hehub_amd: the two batches have different sizes.");
}

// the checks of mult_low_level (ckks/arith.cpp:55-62 -> operator*, rns.h:253-270), member by member; false: not one common shape
template <class Ct> bool mult_args_ok(const std::vector<Ct> &a, const std::vector<Ct> &b, size_t &n, size_t &L, std::vector<u64> &q) {
    size_t nb, Lb;
    std::vector<u64> qb;
    const bool ua = uniform_shape(a, n, L, q), ub = uniform_shape(b, nb, Lb, qb);
    if (!ua || !ub || nb != n || Lb != L) return false;
    for (size_t i = 0; i < a.size(); i++)
        for (int h = 0; h < 2; h++) {
            if (a[i][h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
            if (b[i][h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
        }
    if (q != qb) throw std::invalid_argument("Operands' moduli mismatch.");
    return true;
}

// ckks::mult / bgv mult_low_level + relinearize [+ drop of q_last] on a batch; bgv: t = the common plain modulus
template <class Ct>
std::vector<Ct> mult_batch(const std::vector<Ct> &a, const std::vector<Ct> &b, const RlweKsk &key, bool drop, bool bgv, u64 t, size_t n, size_t L,
                           const std::vector<u64> &q) {
    const size_t B = a.size();
    size_t logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(result_poly(n, L, q, PolyRepForm::value), key, mext);
    if (bgv && L0 != L) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    if (drop && L == 1) throw std::invalid_argument("Unable to drop the only one prime.");
    const size_t Lout = drop ? L - 1 : L;
    std::vector<Ct> out = result_shells<Ct>(B, n, Lout, q);
    for_slices(B, [&](size_t lo, size_t hi) {
        const size_t S = hi - lo;
        DevKey dk(key, L0, n);   // (the key on this slice's device: assembled there on first use, cached per rank)
        Src d1 = Access::batch_in(halves(a, lo, hi), L), d2 = Access::batch_in(halves(b, lo, hi), L);
        Dst dout(S * 2 * Lout * n);
        if (drop) {
            if (bgv) check(hp_dev_bgv_mult_relin_modswitch(cur(), logn, L, mext.data(), t, S, d1.p, d2.p, dk.p(), dout.p));
            else check(hp_dev_ckks_mult_relin_rescale_at(cur(), logn, L, L0, mext.data(), S, d1.p, d2.p, dk.p(), dout.p));
        } else {
            Dst dq(S * 3 * L * n);
            check(hp_dev_mult_low_level(cur(), logn, L, q.data(), S, d1.p, d2.p, dq.p));
            if (bgv) check(hp_dev_bgv_relinearize(cur(), logn, L, mext.data(), 1 /* bgv.h:32 */, S, dq.p, dk.p(), dout.p));
            else check(hp_dev_ckks_relinearize_at(cur(), logn, L, L0, mext.data(), S, dq.p, dk.p(), dout.p));
        }
        bind_slice(out, lo, hi, dout, Lout);
    });
    return out;
}

// rescale_inplace / mod_switch_inplace by one prime on a batch of one shape
template <class Ct> void drop_batch(std::vector<Ct> &cts, bool bgv, u64 t, size_t n, size_t L, const std::vector<u64> &q) {
    const size_t B = cts.size();
    size_t logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    for_slices(B, [&](size_t lo, size_t hi) {
        const size_t S = hi - lo;
        Src din = Access::batch_in(halves(cts, lo, hi), L);
        Dst dout(S * 2 * (L - 1) * n);
        if (bgv) check(hp_dev_bgv_mod_switch(cur(), logn, L, q.data(), t, S, din.p, dout.p));
        else check(hp_dev_ckks_rescale(cur(), logn, L, q.data(), S, din.p, dout.p));
#ifdef HEHUB_AMD_BIND_REFERENCE
        limb_copies_wait();   // (remove_components hands limb blocks back to hehub's pool: their uploads must have happened, see drop_last_prime)
#endif
        std::vector<RnsIntVec *> polys;
        for (size_t i = lo; i < hi; i++)
            for (int h = 0; h < 2; h++) {
                cts[i][h].remove_components();
                polys.push_back(&cts[i][h]);
            }
        Access::bind_many(polys, dout, L - 1);
    });
}

std::vector<ckks::CkksCt> ckks_mult(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key, bool rescale) {
    same_size(a.size(), b.size());
    size_t n, L;
    std::vector<u64> q;
    std::vector<ckks::CkksCt> out;
    if (a.empty()) return out;
    if (!mult_args_ok(a, b, n, L, q)) {   // no common shape: the loop of single calls, with their checks
        for (size_t i = 0; i < a.size(); i++) {
            out.push_back(ckks::mult(a[i], b[i], key));
            if (rescale) ckks::rescale_inplace(out.back());
        }
        return out;
    }
    out = mult_batch(a, b, key, rescale, false, 0, n, L, q);
    for (size_t i = 0; i < a.size(); i++) {
        out[i].scaling_factor = a[i].scaling_factor * b[i].scaling_factor;   // ckks/arith.cpp:61, :68
        if (rescale) out[i].scaling_factor /= q[L - 1];                      // rescaling.cpp:87
    }
    return out;
}

std::vector<bgv::BgvCt> bgv_mult(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key, bool drop) {
    same_size(a.size(), b.size());
    size_t n, L;
    std::vector<u64> q;
    std::vector<bgv::BgvCt> out;
    if (a.empty()) return out;
    bool one_t = true;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].plain_modulus != b[i].plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");   // bgv/arith.cpp:60-62
        one_t = one_t && a[i].plain_modulus == a[0].plain_modulus;
    }
    if (!one_t || !mult_args_ok(a, b, n, L, q)) {
        for (size_t i = 0; i < a.size(); i++) {
            out.push_back(bgv::relinearize(bgv::mult_low_level(a[i], b[i]), key));
            if (drop) bgv::mod_switch_inplace(out.back());
        }
        return out;
    }
    out = mult_batch(a, b, key, drop, true, a[0].plain_modulus, n, L, q);
    for (auto &ct : out) ct.plain_modulus = a[0].plain_modulus;
    return out;
}

std::vector<ckks::CkksCt> ckks_key_switched(const std::vector<ckks::CkksCt> &cts, const RlweKsk &key, bool conj, size_t step) {
    size_t n, L;
    std::vector<u64> q;
    std::vector<ckks::CkksCt> out;
    if (cts.empty()) return out;
    if (!uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) out.push_back(conj ? ckks::conjugate(ct, key) : ckks::rotate(ct, key, step));
        return out;
    }
    for (auto &ct : cts)
        for (int h = 0; h < 2; h++)
            if (ct[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(cts[0][1], key, mext);
    const size_t logn = cts[0][1].log_dimension(), B = cts.size();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    for_slices(B, [&](size_t lo, size_t hi) {
        DevKey dk(key, L0, n);
        Src din = Access::batch_in(halves(cts, lo, hi), L);
        Dst dout((hi - lo) * 2 * L * n);
        if (conj) check(hp_dev_ckks_conjugate_at(cur(), logn, L, L0, mext.data(), hi - lo, din.p, dk.p(), dout.p));
        else check(hp_dev_ckks_rotate_at(cur(), logn, L, L0, mext.data(), hi - lo, step, din.p, dk.p(), dout.p));
        bind_slice(out, lo, hi, dout, L);
    });
    for (size_t i = 0; i < B; i++) out[i].scaling_factor = cts[i].scaling_factor;
    return out;
}

// ckks::rotate(cts[i], *keys[i], steps[i]) for every i as ONE engine call (hp_dev_ckks_rotate_many); a batch that is not of one shape,
// or whose keys were not all made for the same number of moduli, runs as the loop of single calls
std::vector<ckks::CkksCt> ckks_rotate_many(const std::vector<const ckks::CkksCt *> &cts, const std::vector<const RlweKsk *> &keys,
                                           const std::vector<size_t> &steps) {
    same_size(cts.size(), keys.size());
    same_size(cts.size(), steps.size());
    std::vector<ckks::CkksCt> out;
    if (cts.empty()) return out;
    for (const RlweKsk *k : keys)
        if (!k) throw std::invalid_argument("Empty RGSW ciphertext.");
    const size_t n = (*cts[0])[0].dimension(), L = (*cts[0])[0].component_count(), B = cts.size();
    std::vector<u64> q((*cts[0])[0].modulus_vec());
    q.resize(L);
    bool uniform = L > 0 && n >= 2;
    for (size_t i = 0; uniform && i < B; i++) {
        for (int h = 0; h < 2 && uniform; h++) {
            const RnsPolynomial &p = (*cts[i])[h];
            std::vector<u64> m(p.modulus_vec());
            m.resize(L);
            uniform = p.dimension() == n && p.component_count() == L && m == q;
        }
        uniform = uniform && keys[i]->size() == keys[0]->size();
    }
    if (!uniform) {
        for (size_t i = 0; i < B; i++) out.push_back(ckks::rotate(*cts[i], *keys[i], steps[i]));
        return out;
    }
    for (const ckks::CkksCt *ct : cts)
        for (int h = 0; h < 2; h++)
            if ((*ct)[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext, mext_i;
    const size_t L0 = check_ext_prod((*cts[0])[1], *keys[0], mext);
    for (size_t i = 1; i < B; i++)
        if (check_ext_prod((*cts[i])[1], *keys[i], mext_i) != L0 || mext_i != mext) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    const size_t logn = (*cts[0])[1].log_dimension();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    // ONE ciphertext under many keys (src/circuits/linear_algebra.h:123-130) is one operand: it stays on its device, the batch is not cut
    // (a slice elsewhere would drag the vector and every key of the slice over).  Different ciphertexts: contiguous slices, one per rank.
    bool one_ct = true;
    for (size_t i = 1; i < B; i++) one_ct = one_ct && cts[i] == cts[0];
    auto run = [&](size_t lo, size_t hi) {
        std::vector<DevKey> dks;
        dks.reserve(hi - lo);
        std::vector<const u64 *> kp;
        for (size_t i = lo; i < hi; i++) {   // (a key that appears several times is assembled once: the key cache, or the earlier element)
            size_t same = i;
            for (size_t j = lo; j < i && same == i; j++)
                if (keys[j] == keys[i]) same = j;
            if (same < i) { kp.push_back(kp[same - lo]); continue; }
            dks.emplace_back(*keys[i], L0, n);
            kp.push_back(dks.back().p());
        }
        std::vector<const u64 *> polys;   // the operands are read where they are: the same object may appear many times
        std::vector<Src> holds;
        for (size_t i = lo; i < hi; i++)
            for (int h = 0; h < 2; h++) {
                holds.push_back(Access::in((*cts[i])[h], L));
                polys.push_back(holds.back().p);
            }
        Dst dout((hi - lo) * 2 * L * n);
        check(hp_dev_ckks_rotate_many_rows(cur(), logn, L, L0, mext.data(), hi - lo, steps.data() + lo, nullptr, polys.data(), kp.data(), dout.p));
        bind_slice(out, lo, hi, dout, L);
    };
    if (one_ct) {
        OpScope op({Access::home((*cts[0])[0]), Access::home((*cts[0])[1])}, 0);
        run(0, B);
    } else {
        for_slices(B, run);
    }
    for (size_t i = 0; i < B; i++) out[i].scaling_factor = cts[i]->scaling_factor;
    return out;
}

std::vector<ckks::CkksCt> ckks_addsub(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, bool sub) {
    same_size(a.size(), b.size());
    size_t n, L, nb, Lb;
    std::vector<u64> q, qb;
    std::vector<ckks::CkksCt> out;
    if (a.empty()) return out;
    bool fast = uniform_shape(a, n, L, q) && uniform_shape(b, nb, Lb, qb) && nb == n && Lb >= L;   // (b may carry more limbs: rns.cpp:59-72)
    if (fast) {
        qb.resize(L);
        fast = qb == q;
    }
    for (size_t i = 0; i < a.size() && fast; i++)
        for (int h = 0; h < 2; h++) fast = a[i][h].rep_form == b[i][h].rep_form && a[i][h].rep_form == a[0][0].rep_form;
    for (size_t i = 0; i < a.size() && fast; i++) fast = std::abs(a[i].scaling_factor - b[i].scaling_factor) <= std::pow(2.0, -50);
    if (!fast) {   // the loop of single calls throws what hehub throws, where hehub throws it
        for (size_t i = 0; i < a.size(); i++) out.push_back(sub ? ckks::sub(a[i], b[i]) : ckks::add(a[i], b[i]));
        return out;
    }
    const size_t B = a.size();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    for_slices(B, [&](size_t lo, size_t hi) {
        Src da = Access::batch_in(halves(a, lo, hi), L), db = Access::batch_in(halves(b, lo, hi), L);
        Dst dout((hi - lo) * 2 * L * n);
        dev_binary(sub ? Bin::sub : Bin::add, n, L, q.data(), 2 * (hi - lo), da.p, db.p, dout.p);
        bind_slice(out, lo, hi, dout, L);
    });
    for (size_t i = 0; i < B; i++) {
        out[i].scaling_factor = a[i].scaling_factor;
        for (int h = 0; h < 2; h++) out[i][h].rep_form = a[i][h].rep_form;
    }
    return out;
}

} // namespace

std::vector<ckks::CkksCt> mult(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key) { return ckks_mult(a, b, key, false); }
std::vector<ckks::CkksCt> mult_rescale(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key) { return ckks_mult(a, b, key, true); }
std::vector<bgv::BgvCt> mult(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key) { return bgv_mult(a, b, key, false); }
std::vector<bgv::BgvCt> mult_mod_switch(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key) { return bgv_mult(a, b, key, true); }
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &rot_key, size_t step) { return ckks_key_switched(cts, rot_key, false, step); }
std::vector<ckks::CkksCt> conjugate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &conj_key) { return ckks_key_switched(cts, conj_key, true, 0); }
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const std::vector<const RlweKsk *> &rot_keys, const std::vector<size_t> &steps) {
    std::vector<const ckks::CkksCt *> p;
    for (auto &ct : cts) p.push_back(&ct);
    return ckks_rotate_many(p, rot_keys, steps);
}
std::vector<ckks::CkksCt> rotate(const ckks::CkksCt &ct, const std::vector<const RlweKsk *> &rot_keys, const std::vector<size_t> &steps) {
    return ckks_rotate_many(std::vector<const ckks::CkksCt *>(rot_keys.size(), &ct), rot_keys, steps);
}
std::vector<ckks::CkksCt> add(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b) { return ckks_addsub(a, b, false); }
std::vector<ckks::CkksCt> sub(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b) { return ckks_addsub(a, b, true); }

void rescale_inplace(std::vector<ckks::CkksCt> &cts) {
    size_t n, L;
    std::vector<u64> q;
    if (cts.empty()) return;
    for (auto &ct : cts) check_ct_wellformed(ct);   // rescaling.cpp:15-29
    if (!uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) ckks::rescale_inplace(ct);
        return;
    }
    drop_batch(cts, false, 0, n, L, q);
    for (auto &ct : cts) ct.scaling_factor /= q[L - 1];
}

void mod_switch_inplace(std::vector<bgv::BgvCt> &cts) {
    size_t n, L;
    std::vector<u64> q;
    if (cts.empty()) return;
    for (auto &ct : cts) check_ct_wellformed(ct);   // mod_switch.cpp:14-28
    bool one_t = true;
    for (auto &ct : cts) one_t = one_t && ct.plain_modulus == cts[0].plain_modulus;
    if (!one_t || !uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) bgv::mod_switch_inplace(ct);
        return;
    }
    drop_batch(cts, true, cts[0].plain_modulus, n, L, q);
}

} // namespace amd

} // namespace hehub
