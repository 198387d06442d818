// deferred_record.cpp -- the queue of recorded calls: what a recorded call keeps (layer.hpp: PendingOp), placeholders for results that
// have not been computed, operands that live on another device rank than the call.
#include "layer.hpp"

namespace hehub {

namespace amd {

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

bool deferred() { return op_queue().on; }

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

// record a call: returns the placeholder of its result words
// What the engine would refuse when the call RUNS is refused when it is RECORDED: hehub throws at the call (ntt.cpp:26-29,43-47 for a
// modulus the transforms cannot use), and so does the call-by-call mode (check() on the engine call's status).  hp_check_chain builds
// what the first call on the chain builds anyway (tables, constants -- on the rank the call is recorded for); a chain that has been
// accepted once on a rank is remembered, so a recorded call costs one set lookup.
static void validate(const PendingOp &op) {
    if (op.kind == OpKind::Copy || op.mod.empty()) return;   // (a copy of pending words: no arithmetic, no chain)
    bool ntt = false, mont = false;
    switch (op.kind) {
    case OpKind::Relin: case OpKind::KeySwitch: case OpKind::Drop: case OpKind::Transform: ntt = true; break;
    case OpKind::MultLow: case OpKind::PolyMul: mont = true; break;
    default: break;
    }
    if (op.kind == OpKind::KeySwitch && !op.conj && op.step >= ((size_t)1 << 17)) throw std::invalid_argument("rotation step out of range");
    if (op.kind == OpKind::Drop && op.bgv && op.t == 0) throw std::invalid_argument("plain modulus must be positive");
    typedef std::tuple<int, int, size_t, std::vector<u64>> Key;
    static std::set<Key> &accepted = *new std::set<Key>;   // (never destroyed, like the pool)
    Key key(op.rank, (ntt ? 2 : 0) | (mont ? 1 : 0), ntt ? op.logn : 0, op.mod);
    if (accepted.count(key)) return;
    check(hp_check_chain(cur(), ntt ? op.logn : 0, op.mod.data(), op.mod.size(), mont ? 1 : 0));
    accepted.insert(std::move(key));
}

BlockRef record(std::unique_ptr<PendingOp> op) {
    validate(*op);   // (throws what the call-by-call mode throws, before anything about the call has been written down)
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

} // namespace hehub
