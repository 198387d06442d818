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

} // namespace hehub
