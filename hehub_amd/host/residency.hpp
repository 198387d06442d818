// residency.hpp -- struct Access of the own-mirror build: the vector carries its device copy (hehub.hpp: RnsIntVec).  Where the words
// are, uploads, downloads of one limb or all, moves between device ranks, views of batch blocks.  Included by layer.hpp; not a header
// of its own.
#pragma once

namespace hehub {

namespace amd {

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


} // namespace amd

} // namespace hehub
