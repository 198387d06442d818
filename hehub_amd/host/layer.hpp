// layer.hpp -- INTERNAL header of the hehub-compatible host layer (hehub_amd/host/*.cpp); not installed, not part of the interface.
// The layer validates arguments the way the reference does, finds (or puts) the operands' words on the device, calls the engine through
// the C ABI and binds the results to the objects it returns -- no arithmetic on ring elements happens in it.  Its parts:
//
//   engine_lanes.cpp     the process-wide engine: device ranks, lanes, which call runs where, ordering between lanes, error mapping
//   block_pool.cpp       pooled device blocks per rank, synchronous PCIe copies, PCIe accounting
//   residency.hpp/.cpp   own-mirror build: the two copies of a vector (struct Access), the RnsIntVec members, gather of sibling views
//   binding_cache.hpp / binding.cpp   binding build (-DHEHUB_AMD_BIND_REFERENCE): hehub's own host objects, registered limb blocks,
//                        the page-locked arena, the opt-in cache of device copies (struct Access of that build)
//   deferred_record.cpp  recorded calls: the queue, placeholders, operands found on another rank
//   deferred_run.cpp     running the queue: grouping, the fused pipeline, sum chains
//   scheme_calls.cpp     hehub's functions themselves (rns.h operators, mod_arith.h, ntt.h, permutation.h, rlwe.h / rgsw.h, ckks.h, bgv.h)
//   batched_forms.cpp    hehub_amd_ext.hpp: std::vector<Ct> in, one engine call per device rank
//
// Two ways to build it:
//   default                      against hehub.hpp, our own mirror of the reference's types (hehub_amd/host);
//   -DHEHUB_AMD_BIND_REFERENCE   against the reference's OWN headers (-I<hehub>/src): the layer then defines only the functions hehub
//                                defines out of line on the hot path, with hehub's exact signatures, so linking
//                                it ahead of hehub's ntt.cpp / mod_arith.cpp / rns.cpp / rgsw.cpp / rescaling.cpp / mod_switch.cpp /
//                                arith.cpp / permutation.cpp moves that path to the GPU while everything else (sampling, encoding, key
//                                generation, circuits, tests) stays hehub's.  The `ref_tests` make target does exactly that with hehub's
//                                own test suite.
#pragma once

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
#include <set>
#include <tuple>
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
//   * the batched forms cut a batch into parts, one per rank (no collective): an element stays on the rank its operand
//     lives on, elements that live only on the host fill the ranks up to batch / ranks each, contiguously;
//   * keys and tables are replicated per rank on first use (the key cache is keyed by rank);
//   * an operand that lives on another rank than the call is copied over (hp_memcpy_peer_async: one xGMI link) and, when it is a
//     vector, stays there -- counted in TransferStats::peer_copies.  Ranks may share a GPU (HEHUB_AMD_DEVICES=0,0: the one-GPU tests).
// Known limit: a vector has ONE device copy.  An operand that several ranks read (a shared plaintext, a common second factor) is moved
// to the rank of each call that reads it, not replicated: chains that were spread first and then share an operand pass it back and forth
// (peer_copies shows it).  An operand that is resident BEFORE the chains start draws them all to its rank instead (no copies, no
// spreading).  Independent ciphertexts with their own operands -- the loops of hehub's callers -- have neither problem.
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
LaneSet &lane_set();
hp_ctx *cur();                 // the context of the slot the current call runs on (made on first use)
int cur_rank();                // ... and its device rank
hp_ctx *rank_ctx(int rank);    // the root context of a device rank (made on first use)
void flush_all();              // run everything that has been recorded (deferred_run.cpp)

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

using BlockRef = std::shared_ptr<DevBlock>;

extern TransferStats g_stats;
void check(int rc);            // status of an engine call -> the exception hehub throws for it; counts the call
// the current call reads / writes the block: wait for whoever it depends on, leave the call's ticket
void track_read(DevBlock &b);
void track_write(DevBlock &b);
int home_rank(const DevBlock &b);   // the device rank a block's words live on
void settled(DevBlock &b);     // after a host synchronisation of the current lane that followed track_write: every earlier user has finished
BlockRef alloc_block(size_t words);

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
        if (one && force_rank <= 0) slot = 0;   // (a forced rank > 0 with one rank in use: a look at words left on a rank that went out of use)
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
        try {
            (void)cur();   // (makes the slot's context on first use: may throw -- no engine on that device)
        } catch (...) {
            S.cur = saved_;
            S.depth--;
            throw;
        }
        if (!nested) S.last = slot;
        S.v[slot].ticket++;
    }
    ~OpScope() {
        LaneSet &S = lane_set();
        if (--S.depth > 0 && S.cur != saved_) {
            // a forced scope inside another call, on another slot: that call goes on where it was -- under a NEW ticket.  While the
            // inner scope ran, its slot may have ordered itself behind the outer call's slot (order_after: "seen up to ticket T", T the
            // outer call's OPEN ticket); what the outer call enqueues from here on must not pass for seen -- a block it still uses
            // would look free to the inner slot's next call (found by the random-program fuzz over device ranks: 1 program in 129)
            S.cur = saved_;
            S.v[saved_].ticket++;
        }
    }
    OpScope(const OpScope &) = delete;

private:
    int saved_ = 0;
};
void h2d(u64 *dst, const u64 *src, size_t words);   // synchronous on the current lane, counted
void d2h(u64 *dst, const u64 *src, size_t words);
#ifdef HEHUB_AMD_BIND_REFERENCE
// hehub's own limbs (binding.cpp): registered blocks cross PCIe by DMA / by one kernel per polynomial, small ones through a page-locked arena
void limb_copy_h2d(u64 *dst, const u64 *src, size_t words);
void limb_copy_d2h(u64 *dst, const u64 *src, size_t words);
void limb_copies_wait();
void poly_copy_h2d(u64 *dst, const RnsIntVec &v, size_t limbs, size_t n);
void poly_copy_d2h(RnsIntVec &v, const u64 *src, size_t limbs, size_t n);
bool cache_get(const RnsIntVec &v, size_t limbs, BlockRef &blk, size_t &off);
void cache_put(const RnsIntVec &v, size_t limbs, const BlockRef &blk, size_t off);
#else
// the mirror's limbs (residency.cpp): separate host vectors, ONE copy through a page-locked staging buffer
void h2d_limbs(u64 *dst, const std::vector<std::vector<u64>> &limbs, size_t count, size_t n);
void d2h_limbs(std::vector<std::vector<u64>> &limbs, const u64 *src, size_t count, size_t n);
unsigned long long next_stamp();
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
// Results are word for word those of the eager calls, exceptions included: record() asks the engine whether it accepts the call's modulus
// chain (hp_check_chain) before anything is written down, so a modulus the transforms reject throws at the call.  What differs: a failure
// of the DEVICE (a HIP error, out of memory) surfaces when the queue runs, not at the call that recorded it.
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
};OpQueue &op_queue();
BlockRef record(std::unique_ptr<PendingOp> op);   // record a call: returns the placeholder of its result words
bool deferred();
u64 *words_of(const BlockRef &b);                 // device address of a block's words; a placeholder is resolved by running the queue
void peer_fetch(u64 *dst, const BlockRef &b, size_t off, size_t words);
Src here(const BlockRef &b, size_t off, size_t words);
const u64 *key_here(const BlockRef &key);

} // namespace amd

} // namespace hehub

#ifdef HEHUB_AMD_BIND_REFERENCE
#include "binding_cache.hpp"
#else
#include "residency.hpp"
#endif

namespace hehub {

namespace amd {
Src gather(std::initializer_list<const RnsIntVec *> polys, size_t limbs);
} // namespace amd

using amd::Access;
using amd::check;
using amd::OpScope;
using amd::Dst;
using amd::Src;

// what the scheme-level calls and their batched forms share (scheme_calls.cpp)
namespace detail {
RnsPolynomial result_poly(size_t n, size_t limbs, const std::vector<u64> &moduli, PolyRepForm form);
enum class Bin { add, sub, mul };
void dev_binary(Bin op, size_t n, size_t L, const u64 *m, size_t batch, const u64 *a, const u64 *b, u64 *out);
void check_ct_wellformed(const RlweCt &ct);
bool extensions_on();
size_t check_ext_prod(const RlwePt &pt, const RgswCt &rgsw, std::vector<u64> &extended_moduli);
// Device copy of a key-switching key: u64[L][2][L+1][N], one block on the CURRENT device rank, assembled from the key's 2L polynomials
// and cached per rank (scheme_calls.cpp)
class DevKey {
public:
    DevKey(const RgswCt &rgsw, size_t L, size_t n);
    const u64 *p() const { return own_->p; }
    const amd::BlockRef &block() const { return own_; }

private:
    static void assemble(u64 *dst, const RgswCt &rgsw, size_t L, size_t n);
    amd::BlockRef own_;
};
} // namespace detail

} // namespace hehub
