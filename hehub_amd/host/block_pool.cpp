// block_pool.cpp -- device blocks out of a per-rank, per-size free list, and the synchronous copies between host and device.
#include "layer.hpp"

namespace hehub {

namespace amd {

namespace {

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
} // namespace

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

} // namespace amd

} // namespace hehub
