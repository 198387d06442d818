// binding.cpp -- the binding build only (-DHEHUB_AMD_BIND_REFERENCE): hehub's own objects are host memory.  Their limbs cross PCIe from
// blocks registered with the driver (or through a page-locked arena), and an opt-in cache remembers which device block holds the words
// of which host polynomial.  (struct Access of this build: binding_cache.hpp.)
#include "layer.hpp"

#ifdef HEHUB_AMD_BIND_REFERENCE

namespace hehub {

namespace amd {

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
void poly_copy_h2d(u64 *dst, const RnsIntVec &v, size_t limbs, size_t n) {
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
void poly_copy_d2h(RnsIntVec &v, const u64 *src, size_t limbs, size_t n) {
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
} // namespace

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



} // namespace amd

} // namespace hehub

#endif
