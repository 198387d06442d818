// deferred_run.cpp -- running the queue of recorded calls: calls with one signature whose operands are ready run as ONE batched engine
// call on the device rank they were recorded for; mult + relinearize + drop triples as the engine's one-call pipeline; chains of sums as
// one pass.  Results are word for word those of the single calls.
#include "layer.hpp"

namespace hehub {

namespace amd {

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

} // namespace amd

} // namespace hehub
