// binding_cache.hpp -- struct Access of the binding build: hehub's own types, host objects are hehub's, the device side is a cache
// (binding.cpp).  Included by layer.hpp; not a header of its own.
#pragma once

namespace hehub {

namespace amd {

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

} // namespace amd

} // namespace hehub
