// hp_node.cpp -- several GPUs of one node behind one handle (include/hehub_amd.h, "node").
//
// hehub itself is single-threaded and knows nothing about devices; a host application that holds a batch of
// ciphertexts uses all GPUs of a node through this layer without Python or torch.distributed:
//
//   * batch-sharded mode (the throughput mode): the batch is cut into contiguous per-rank slices
//     (hp_node_slice), every rank runs the SAME single-GPU entry point on its slice, keys and tables are replicated,
//     no data-path exchange.  One worker thread per rank feeds its own hp_ctx (own stream), so the ranks' host-side
//     copies and launches overlap.
//   * limb-sharded mode (the "latency" mode; north star: "all-gather ... only for the key-switch accumulation"): ONE
//     batch is processed by all ranks, cut by OUTPUT MODULUS; rank r owns a contiguous range of q_0..q_{L-1}, p.  Every
//     sum over the digits for an output modulus is formed on one GPU in the reference's order (rgsw.cpp:126-149), so
//     results stay bit-identical.  The exchanges are DIRECT PEER WRITES: the owner of a limb copies it straight into
//     every peer's buffer (hipMemcpy2DAsync between devices with peer access enabled -- each shard crosses exactly one
//     xGMI link, no ring, no padding, no staging), ordered by HIP events; buffers are allocated once per plan.
//
// Ranks may share a device (devices = {0, 0, 0}): that is how the tests run every code path of this file on a one-GPU box.
#include "hp_ctx.h"

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>

#include <rccl/rccl.h>   // types and prototypes only: the library itself is loaded on demand (hp_node_set_transport), never linked

using namespace hpi;

// ---- RCCL, loaded on demand ---------------------------------------------------------------------------------------------------------
// north star: "RCCL all-gather over xGMI ... for the key-switch accumulation".  The limb-sharded plan below can run its exchanges as
// RCCL collectives (HP_TRANSPORT_RCCL) instead of direct peer writes; the engine library does not link librccl -- a process that never
// asks for the collective never loads it.
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;   // when lib == nullptr: what went wrong
};
Rccl &rccl() {
    static Rccl &r = *[] {
        Rccl *x = new Rccl;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x->lib) break;
        }
        if (!x->lib) {
            const char *e = dlerror();
            x->why = std::string("librccl.so could not be loaded: ") + (e ? e : "?");
            return x;
        }
        x->CommInitAll = (decltype(x->CommInitAll))dlsym(x->lib, "ncclCommInitAll");
        x->CommDestroy = (decltype(x->CommDestroy))dlsym(x->lib, "ncclCommDestroy");
        x->AllGather = (decltype(x->AllGather))dlsym(x->lib, "ncclAllGather");
        x->Broadcast = (decltype(x->Broadcast))dlsym(x->lib, "ncclBroadcast");
        x->GetErrorString = (decltype(x->GetErrorString))dlsym(x->lib, "ncclGetErrorString");
        if (!x->CommInitAll || !x->CommDestroy || !x->AllGather || !x->Broadcast || !x->GetErrorString) {
            x->why = "librccl.so lacks one of ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclBroadcast / ncclGetErrorString";
            x->lib = nullptr;
        }
        return x;
    }();
    return r;
}
} // namespace

namespace {

// a reusable barrier for the worker threads (hipStreamWaitEvent only orders against events that have ALREADY been recorded
// on the host side, so "everybody has recorded" is a host-side rendezvous)
class HostBarrier {
public:
    explicit HostBarrier(size_t n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const size_t gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            gen_++;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen; });
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    size_t n_, count_ = 0, gen_ = 0;
};

// per-rank staging of a host-resident batch: device buffers grow on demand and are kept with the node
struct Staging {
    void *d[3] = {nullptr, nullptr, nullptr};
    size_t bytes[3] = {0, 0, 0};
};

// ---- NUMA placement -------------------------------------------------------------------------------------------------------------
// An 8-GPU MI355X node has two sockets with four GPUs each; a rank's host thread (launches, staging copies of host-resident batches,
// page-locked buffers it allocates) belongs on the socket its GPU hangs off: PCI address of the HIP device -> sysfs numa_node ->
// the node's cpulist.  -1 / empty where the platform does not say (one-socket boxes, containers without sysfs).
int device_numa(int device, std::string *cpulist) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = std::fopen(path, "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node >= 0 && cpulist) {
        std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        if (FILE *f = std::fopen(path, "r")) {
            char buf[1024] = {0};
            if (std::fgets(buf, sizeof(buf), f)) {
                cpulist->assign(buf);
                while (!cpulist->empty() && (cpulist->back() == '\n' || cpulist->back() == ' ')) cpulist->pop_back();
            }
            std::fclose(f);
        }
    }
    return node;
}
// "0-3,8,10-11" -> cpu set, intersected with what the process may use; the number of CPUs in the result
int parse_cpulist(const std::string &list, cpu_set_t *out) {
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
    CPU_ZERO(out);
    const char *p = list.c_str();
    int count = 0;
    while (*p) {
        char *end = nullptr;
        long a = std::strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, out); count++; }
        p = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return count;
}

struct Worker {
    int numa_node = -1, cpus_bound = 0;   // where the thread was put (hp_node_placement)
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, quit = false, started = false;   // started: the placement fields above are final
    int rc = HP_OK;
    std::string msg;   // the failing call's message, taken on the worker thread itself (its thread-local slot is the right one)
};

// status of a rank whose own calls all succeeded but which stopped because a peer failed (limb-sharded mode): reported only
// when no rank has a failure of its own
constexpr int HP_PEER_FAILED = 1000;

} // namespace

struct hp_node {
    std::vector<int> devices;
    std::vector<char> peer;   // [a * size + b] != 0: rank a writes rank b's device memory directly (same device, or peer access enabled)
    bool no_peer = false;     // HP_NODE_NO_PEER: pretend no pair has direct access (tests: the staged exchange on a one-GPU box)
    std::vector<hp_ctx *> ctx;
    std::vector<std::unique_ptr<Worker>> workers;
    std::unique_ptr<HostBarrier> barrier;
    std::vector<Staging> staging;
    std::mutex mu;        // one node-level call at a time
    std::string err;
    // how the limb-sharded plans made from now on exchange their limbs (hp_node_set_transport)
    int transport = HP_TRANSPORT_PEER;
    std::vector<ncclComm_t> comms;   // HP_TRANSPORT_RCCL: one communicator rank per node rank (ncclCommInitAll over the node's devices)
};

namespace {

void worker_loop(Worker *w, int device) {
    // the rank's thread runs on the socket of its GPU (HP_NODE_NO_AFFINITY: leave it where the scheduler puts it)
    int numa = -1, bound = 0;
    if (!getenv("HP_NODE_NO_AFFINITY")) {
        std::string cpus;
        numa = device_numa(device, &cpus);
        cpu_set_t set;
        if (numa >= 0 && !cpus.empty()) {
            const int n = parse_cpulist(cpus, &set);
            if (n > 0 && pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0) bound = n;
        }
    }
    {   // hp_node_create waits for this: hp_node_placement never reads a half-written record
        std::lock_guard<std::mutex> lk(w->mu);
        w->numa_node = numa;
        w->cpus_bound = bound;
        w->started = true;
    }
    w->cv.notify_all();
    for (;;) {
        std::function<int()> task;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->has_task || w->quit; });
            if (w->quit) return;
            task = std::move(w->task);
            w->has_task = false;
        }
        int rc;
        try {
            rc = task();
        } catch (...) {
            rc = HP_ELOGIC;
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->rc = rc;
            w->done = true;
        }
        w->cv.notify_all();
    }
}

// run fn(rank) on every rank's worker thread at once; the first non-zero status (lowest rank) is returned and its message kept
int run_all(hp_node *node, const std::function<int(size_t)> &fn) {
    const size_t n = node->ctx.size();
    for (size_t r = 0; r < n; r++) {
        Worker *w = node->workers[r].get();
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->msg.clear();
            w->task = [&fn, r, node, w] {
                const int rc = fn(r);
                if (rc != HP_OK && rc != HP_PEER_FAILED) w->msg = hp_last_error(node->ctx[r]);
                return rc;
            };
            w->has_task = true;
            w->done = false;
        }
        w->cv.notify_all();
    }
    int first = HP_OK;
    bool peer_failed = false;
    for (size_t r = 0; r < n; r++) {
        Worker *w = node->workers[r].get();
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done; });
        if (w->rc == HP_PEER_FAILED) peer_failed = true;
        else if (w->rc != HP_OK && first == HP_OK) {
            first = w->rc;
            node->err = "rank " + std::to_string(r) + ": " + w->msg;
        }
    }
    if (first == HP_OK && peer_failed) {   // (cannot happen: somebody's own failure is what stops the others)
        first = HP_ELOGIC;
        node->err = "a rank stopped because a peer failed, but no rank reported a failure of its own";
    }
    return first;
}

int node_fail(hp_node *node, int code, const std::string &msg) {
    node->err = msg;
    return code;
}

void slice_of(size_t total, size_t world, size_t rank, size_t *lo, size_t *hi) {
    const size_t base = total / world, extra = total % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

} // namespace

namespace {

int stage_reserve(hp_ctx *ctx, Staging &s, int slot, size_t bytes) {
    if (bytes <= s.bytes[slot]) return HP_OK;
    if (s.d[slot]) {
        int rc = hp_dev_free(ctx, s.d[slot]);
        if (rc) return rc;
        s.d[slot] = nullptr;
        s.bytes[slot] = 0;
    }
    int rc = hp_dev_alloc(ctx, bytes, &s.d[slot]);
    if (rc) return rc;
    s.bytes[slot] = bytes;
    return HP_OK;
}

} // namespace

extern "C" {

int hp_node_create(const int *devices, size_t count, hp_node **out) {
    if (!devices || !out || count == 0 || count > 64) return HP_EINVAL;
    hp_node *node = new (std::nothrow) hp_node();
    if (!node) return HP_ENOMEM;
    int prev_dev = -1;
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev_dev};   // the caller's device stays current
    node->devices.assign(devices, devices + count);
    for (size_t r = 0; r < count; r++) {
        hp_ctx *c = nullptr;
        int rc = hp_ctx_create(devices[r], &c);
        if (rc != HP_OK) {
            for (hp_ctx *p : node->ctx) hp_ctx_destroy(p);
            delete node;
            return rc;
        }
        node->ctx.push_back(c);
    }
    // peer access between every pair of distinct devices (direct writes over xGMI in the limb-sharded mode); what was granted is
    // kept per pair (hp_node_peer_matrix) and the limb-sharded plan stages through page-locked host memory where it was not
    node->no_peer = getenv("HP_NODE_NO_PEER") != nullptr;
    node->peer.assign(count * count, 0);
    for (size_t a = 0; a < count; a++)
        for (size_t b = 0; b < count; b++) {
            if (node->no_peer) { node->peer[a * count + b] = (a == b); continue; }
            if (devices[a] == devices[b]) { node->peer[a * count + b] = 1; continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                (void)hipSetDevice(devices[a]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                node->peer[a * count + b] = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
            }
            (void)hipGetLastError();
        }
    node->staging.resize(count);
    node->barrier.reset(new HostBarrier(count));
    for (size_t r = 0; r < count; r++) {
        node->workers.emplace_back(new Worker());
        Worker *w = node->workers.back().get();
        w->th = std::thread(worker_loop, w, devices[r]);
    }
    for (auto &w : node->workers) {   // every rank's thread is where it will stay (and says where) before the node is handed out
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->started; });
    }
    *out = node;
    return HP_OK;
}

void hp_node_destroy(hp_node *node) {
    if (!node) return;
    for (auto &w : node->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        w->th.join();
    }
    for (size_t r = 0; r < node->staging.size(); r++)
        for (int s = 0; s < 3; s++)
            if (node->staging[r].d[s]) (void)hp_dev_free(node->ctx[r], node->staging[r].d[s]);
    for (ncclComm_t c : node->comms)
        if (c) (void)rccl().CommDestroy(c);
    for (hp_ctx *c : node->ctx) hp_ctx_destroy(c);
    delete node;
}

int hp_node_get_transport(const hp_node *node) { return node ? node->transport : HP_EINVAL; }
int hp_node_set_transport(hp_node *node, int transport) {
    if (!node) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    if (transport != HP_TRANSPORT_PEER && transport != HP_TRANSPORT_PACKED && transport != HP_TRANSPORT_RCCL)
        return node_fail(node, HP_EINVAL, "transport: HP_TRANSPORT_PEER, HP_TRANSPORT_PACKED or HP_TRANSPORT_RCCL");
    if (transport == HP_TRANSPORT_RCCL && node->comms.empty()) {
        Rccl &R = rccl();
        if (!R.lib) return node_fail(node, HP_EUNSUPPORTED, R.why);
        // one communicator rank per node rank; RCCL refuses two ranks of one communicator on one device, so ranks that share a GPU
        // (the one-GPU tests) cannot use it -- HP_TRANSPORT_PACKED moves the same packed buffers by plain copies there
        for (size_t a = 0; a < node->devices.size(); a++)
            for (size_t b = a + 1; b < node->devices.size(); b++)
                if (node->devices[a] == node->devices[b])
                    return node_fail(node, HP_EUNSUPPORTED, "HP_TRANSPORT_RCCL needs every rank on its own device (ranks " + std::to_string(a) + " and " +
                                                                std::to_string(b) + " share one); HP_TRANSPORT_PACKED exchanges the same buffers by copies");
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        std::vector<ncclComm_t> comms(node->devices.size(), nullptr);
        const ncclResult_t e = R.CommInitAll(comms.data(), (int)node->devices.size(), node->devices.data());
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != ncclSuccess) return node_fail(node, HP_EHIP, std::string("ncclCommInitAll: ") + R.GetErrorString(e));
        node->comms = comms;
    }
    node->transport = transport;
    return HP_OK;
}

size_t hp_node_size(const hp_node *node) { return node ? node->ctx.size() : 0; }
int hp_node_placement(const hp_node *node, size_t rank, int *numa_node, int *cpus_bound) {
    if (!node || rank >= node->workers.size()) return HP_EINVAL;
    if (numa_node) *numa_node = node->workers[rank]->numa_node;
    if (cpus_bound) *cpus_bound = node->workers[rank]->cpus_bound;
    return HP_OK;
}
int hp_device_numa(int device, int *numa_node, char *cpulist, size_t cap) {
    std::string cpus;
    const int node = device_numa(device, &cpus);
    if (numa_node) *numa_node = node;
    if (cpulist && cap) {
        std::snprintf(cpulist, cap, "%s", cpus.c_str());
    }
    return HP_OK;
}
hp_ctx *hp_node_ctx(hp_node *node, size_t rank) { return (node && rank < node->ctx.size()) ? node->ctx[rank] : nullptr; }
const char *hp_node_last_error(hp_node *node) { return node ? node->err.c_str() : "null node"; }
// parity level of every rank's context (hp_ctx_set_parity_level): the batch-sharded entry points then return canonical residues, and
// so does the limb-sharded mode (its stages work on rows the engine wrote itself: the _strict limb-range stages follow the level)
int hp_node_set_parity_level(hp_node *node, int level) {
    if (!node) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    for (hp_ctx *c : node->ctx) {
        int rc = hp_ctx_set_parity_level(c, level);
        if (rc) return node_fail(node, rc, hp_last_error(c));
    }
    return HP_OK;
}

int hp_node_peer_matrix(const hp_node *node, int *matrix) {
    if (!node || !matrix) return HP_EINVAL;
    for (size_t i = 0; i < node->peer.size(); i++) matrix[i] = node->peer[i];
    return HP_OK;
}

int hp_node_slice(const hp_node *node, size_t total, size_t rank, size_t *lo, size_t *hi) {
    if (!node || !lo || !hi || rank >= node->ctx.size()) return HP_EINVAL;
    slice_of(total, node->ctx.size(), rank, lo, hi);
    return HP_OK;
}

int hp_node_sync(hp_node *node) {
    if (!node) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    return run_all(node, [&](size_t r) { return hp_sync(node->ctx[r]); });
}

int hp_node_replicate(hp_node *node, const uint64_t *h_words, size_t words, uint64_t **d_copies) {
    if (!node || !h_words || !d_copies || words == 0) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    for (size_t r = 0; r < node->ctx.size(); r++) d_copies[r] = nullptr;
    return run_all(node, [&](size_t r) {
        int rc = hp_dev_alloc(node->ctx[r], words * 8, (void **)&d_copies[r]);
        if (rc) return rc;
        return hp_memcpy_h2d(node->ctx[r], d_copies[r], h_words, words * 8);
    });
}

int hp_node_free_replicas(hp_node *node, uint64_t **d_copies) {
    if (!node || !d_copies) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    return run_all(node, [&](size_t r) {
        if (!d_copies[r]) return (int)HP_OK;
        int rc = hp_dev_free(node->ctx[r], d_copies[r]);
        d_copies[r] = nullptr;
        return rc;
    });
}

// ---- batch-sharded mode ---------------------------------------------------------------------------------------
// device-resident operands: rank r works on counts[r] items behind its own pointers
static int node_dev_mult(hp_node *node, bool bgv, size_t logn, size_t L, const uint64_t *mext, uint64_t t, const size_t *counts,
                         const uint64_t *const *d_ct1, const uint64_t *const *d_ct2, uint64_t *const *d_key, uint64_t *const *d_out) {
    if (!node || !mext || !counts || !d_ct1 || !d_ct2 || !d_key || !d_out) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(node->mu);
    return run_all(node, [&](size_t r) {
        if (counts[r] == 0) return (int)HP_OK;
        hp_ctx *c = node->ctx[r];
        return bgv ? hp_dev_bgv_mult_relin_modswitch(c, logn, L, mext, t, counts[r], d_ct1[r], d_ct2[r], d_key[r], d_out[r])
                   : hp_dev_ckks_mult_relin_rescale(c, logn, L, mext, counts[r], d_ct1[r], d_ct2[r], d_key[r], d_out[r]);
    });
}
int hp_node_dev_ckks_mult_relin_rescale(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, const size_t *counts,
                                        const uint64_t *const *d_ct1, const uint64_t *const *d_ct2, uint64_t *const *d_key,
                                        uint64_t *const *d_out) {
    return node_dev_mult(node, false, logn, L, moduli_ext, 0, counts, d_ct1, d_ct2, d_key, d_out);
}
int hp_node_dev_bgv_mult_relin_modswitch(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus,
                                         const size_t *counts, const uint64_t *const *d_ct1, const uint64_t *const *d_ct2,
                                         uint64_t *const *d_key, uint64_t *const *d_out) {
    return node_dev_mult(node, true, logn, L, moduli_ext, plain_modulus, counts, d_ct1, d_ct2, d_key, d_out);
}

// host-resident operands (what a hehub application holds): every rank stages its slice in, computes, stages it out
static int node_host_mult(hp_node *node, bool bgv, size_t logn, size_t L, const uint64_t *mext, uint64_t t, size_t batch,
                          const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key, uint64_t *h_out) {
    if (!node || !mext || !h_ct1 || !h_ct2 || !d_key || !h_out) return HP_EINVAL;
    if (logn < 1 || logn > 16 || L < 2) return node_fail(node, HP_EINVAL, "invalid shape");
    std::lock_guard<std::mutex> lk(node->mu);
    std::vector<Staging> &st = node->staging;
    const size_t n = (size_t)1 << logn, in_words = 2 * L * n, out_words = 2 * (L - 1) * n;
    return run_all(node, [&](size_t r) {
        size_t lo, hi;
        slice_of(batch, node->ctx.size(), r, &lo, &hi);
        const size_t cnt = hi - lo;
        if (cnt == 0) return (int)HP_OK;
        hp_ctx *c = node->ctx[r];
        int rc;
        if ((rc = stage_reserve(c, st[r], 0, cnt * in_words * 8))) return rc;
        if ((rc = stage_reserve(c, st[r], 1, cnt * in_words * 8))) return rc;
        if ((rc = stage_reserve(c, st[r], 2, cnt * out_words * 8))) return rc;
        uint64_t *a = (uint64_t *)st[r].d[0], *b = (uint64_t *)st[r].d[1], *o = (uint64_t *)st[r].d[2];
        if ((rc = hp_memcpy_h2d(c, a, h_ct1 + lo * in_words, cnt * in_words * 8))) return rc;
        if ((rc = hp_memcpy_h2d(c, b, h_ct2 + lo * in_words, cnt * in_words * 8))) return rc;
        rc = bgv ? hp_dev_bgv_mult_relin_modswitch(c, logn, L, mext, t, cnt, a, b, d_key[r], o)
                 : hp_dev_ckks_mult_relin_rescale(c, logn, L, mext, cnt, a, b, d_key[r], o);
        if (rc) return rc;
        return hp_memcpy_d2h(c, h_out + lo * out_words, o, cnt * out_words * 8);
    });
}
int hp_node_ckks_mult_relin_rescale(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                                    const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key, uint64_t *h_out) {
    return node_host_mult(node, false, logn, L, moduli_ext, 0, batch, h_ct1, h_ct2, d_key, h_out);
}
int hp_node_bgv_mult_relin_modswitch(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus,
                                     size_t batch, const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key,
                                     uint64_t *h_out) {
    return node_host_mult(node, true, logn, L, moduli_ext, plain_modulus, batch, h_ct1, h_ct2, d_key, h_out);
}

// ntt.h:41-51 / :72-92 on a host-resident batch u64[batch][L][N], in place, sliced over the ranks
int hp_node_ntt(hp_node *node, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *h_x, int inverse, int strict) {
    if (!node || !moduli || !h_x) return HP_EINVAL;
    if (logn < 1 || logn > 16 || L < 1) return node_fail(node, HP_EINVAL, "invalid shape");
    std::lock_guard<std::mutex> lk(node->mu);
    std::vector<Staging> &st = node->staging;
    const size_t words = L * ((size_t)1 << logn);
    return run_all(node, [&](size_t r) {
        size_t lo, hi;
        slice_of(batch, node->ctx.size(), r, &lo, &hi);
        const size_t cnt = hi - lo;
        if (cnt == 0) return (int)HP_OK;
        hp_ctx *c = node->ctx[r];
        int rc;
        if ((rc = stage_reserve(c, st[r], 0, cnt * words * 8))) return rc;
        uint64_t *d = (uint64_t *)st[r].d[0];
        if ((rc = hp_memcpy_h2d(c, d, h_x + lo * words, cnt * words * 8))) return rc;
        rc = inverse ? hp_dev_intt(c, logn, L, moduli, cnt, d, strict) : hp_dev_ntt(c, logn, L, moduli, cnt, d);
        if (rc) return rc;
        return hp_memcpy_d2h(c, h_x + lo * words, d, cnt * words * 8);
    });
}

} // extern "C"

// ---- limb-sharded mode ------------------------------------------------------------------------------------------
struct hp_node_sharded {
    hp_node *node = nullptr;
    size_t logn = 0, L = 0, batch = 0;
    uint64_t t = 0;                        // 0: CKKS, else the BGV plain modulus
    std::vector<uint64_t> mext;
    std::vector<std::pair<size_t, size_t>> own;   // [k0, k1) of the L+1 extended moduli per rank
    struct Rank {
        uint64_t *ct1 = nullptr, *ct2 = nullptr, *quad = nullptr, *coef = nullptr, *ks = nullptr, *c_p = nullptr, *relin = nullptr,
                 *c_q = nullptr, *out = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // coef sent, c_p sent, c_q sent, result sent
        // page-locked host staging per exchange, allocated only for a rank that has a peer it cannot write directly
        uint64_t *hstage[4] = {nullptr, nullptr, nullptr, nullptr};
        // HP_TRANSPORT_PACKED / _RCCL: the two all-gather shaped exchanges (0: coefficient limbs, 3: result limbs) go through a packed
        // send buffer u64[rows][pad][N] (pad = the largest number of limbs any rank owns) and a receive buffer u64[W][rows][pad][N]
        uint64_t *pack[2] = {nullptr, nullptr}, *gath[2] = {nullptr, nullptr};
    };
    std::vector<Rank> rk;
    int transport = HP_TRANSPORT_PEER;   // the node's transport when the plan was made
    size_t pad[2] = {0, 0};              // padded limbs per rank of exchange 0 / 3
};

namespace {

size_t owner_of(const hp_node_sharded *p, size_t k) {
    for (size_t r = 0; r < p->own.size(); r++)
        if (p->own[r].first <= k && k < p->own[r].second) return r;
    return 0;
}

// What one rank sends in one exchange: `rows` rows of `row_limbs` limbs, of which the limbs [lo, hi) travel (a contiguous
// buffer is one row of one "limb" of `n` words).  Source and destinations have the same layout.
struct Block {
    size_t rows = 0, row_limbs = 1, lo = 0, hi = 0, n = 0;
    size_t pitch() const { return row_limbs * n * 8; }
    size_t width() const { return (hi - lo) * n * 8; }
    size_t bytes() const { return rows * width(); }
    bool empty() const { return rows == 0 || hi <= lo; }
};

// the four exchanges of one multiplication as seen from rank r (what r sends)
Block block_of(const hp_node_sharded *p, size_t e, size_t r) {
    const size_t L = p->L, B = p->batch, n = (size_t)1 << p->logn;
    const size_t k0 = p->own[r].first, k1 = p->own[r].second;
    Block b;
    b.n = n;
    switch (e) {
    case 0: b.rows = B; b.row_limbs = L; b.lo = std::min(k0, L); b.hi = std::min(k1, L); break;                  // coefficient limbs
    case 1: if (r == owner_of(p, L)) { b.rows = 1; b.hi = 1; b.n = 2 * B * n; } break;                             // c_p
    case 2: if (r == owner_of(p, L - 1)) { b.rows = 1; b.hi = 1; b.n = 2 * B * n; } break;                         // c_q
    default: b.rows = 2 * B; b.row_limbs = L - 1; b.lo = std::min(k0, L - 1); b.hi = std::min(k1, L - 1); break;   // result limbs
    }
    return b;
}

bool direct(const hp_node *node, size_t from, size_t to) { return node->peer[from * node->ctx.size() + to] != 0; }

// Sender side of an exchange, on rank r's stream: a direct peer write into every destination it may write (one xGMI link per
// shard), ONE copy of the block into its page-locked staging buffer for all the others (they fetch it after the rendezvous).
int send_block(hp_node_sharded *p, size_t r, size_t e, hipStream_t s, const uint64_t *src, const std::function<uint64_t *(size_t)> &dst_of,
               const std::function<bool(size_t)> &is_dest) {
    hp_node *node = p->node;
    hp_ctx *c = node->ctx[r];
    const Block b = block_of(p, e, r);
    if (b.empty()) return HP_OK;
    bool need_stage = false;
    for (size_t d = 0; d < node->ctx.size(); d++) {
        if (d == r || !is_dest(d)) continue;
        if (!direct(node, r, d)) { need_stage = true; continue; }
        const hipError_t er = hipMemcpy2DAsync((char *)dst_of(d) + b.lo * b.n * 8, b.pitch(), (const char *)src + b.lo * b.n * 8, b.pitch(),
                                               b.width(), b.rows, hipMemcpyDeviceToDevice, s);
        if (er != hipSuccess) return chk_local(c, er, "peer copy of owned limbs");
    }
    if (need_stage) {
        if (!p->rk[r].hstage[e]) return fail_local(c, HP_ELOGIC, "limb-sharded plan has no staging buffer for a peer without direct access");
        const hipError_t er = hipMemcpy2DAsync(p->rk[r].hstage[e], b.width(), (const char *)src + b.lo * b.n * 8, b.pitch(), b.width(), b.rows,
                                               hipMemcpyDeviceToHost, s);
        if (er != hipSuccess) return chk_local(c, er, "staging copy of owned limbs (no peer access)");
    }
    return HP_OK;
}

// Receiver side, on rank r's stream, after the rendezvous: wait for the sender's event; where the sender could not write
// here directly, fetch its block from its staging buffer.
int recv_block(hp_node_sharded *p, size_t r, size_t e, hipStream_t s, size_t from, uint64_t *dst) {
    hp_node *node = p->node;
    hp_ctx *c = node->ctx[r];
    const Block b = block_of(p, e, from);
    if (b.empty()) return HP_OK;
    hipError_t er = hipStreamWaitEvent(s, p->rk[from].ev[e], 0);
    if (er != hipSuccess) return chk_local(c, er, "wait for a peer's exchange");
    if (direct(node, from, r)) return HP_OK;
    er = hipMemcpy2DAsync((char *)dst + b.lo * b.n * 8, b.pitch(), p->rk[from].hstage[e], b.width(), b.width(), b.rows, hipMemcpyHostToDevice, s);
    return er == hipSuccess ? (int)HP_OK : chk_local(c, er, "fetch of a peer's staged limbs (no peer access)");
}

void free_plan(hp_node_sharded *p) {
    if (!p) return;
    for (size_t r = 0; r < p->rk.size(); r++) {
        hp_ctx *c = p->node->ctx[r];
        auto &R = p->rk[r];
        for (uint64_t *b : {R.ct1, R.ct2, R.quad, R.coef, R.ks, R.c_p, R.relin, R.c_q, R.out, R.pack[0], R.pack[1], R.gath[0], R.gath[1]})
            if (b) (void)hp_dev_free(c, b);
        for (uint64_t *h : R.hstage)
            if (h) (void)hp_host_free(c, h);
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(p->node->devices[r]);
        for (hipEvent_t e : R.ev)
            if (e) (void)hipEventDestroy(e);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    delete p;
}

// One multiplication, every rank in lockstep on its worker thread.  d_in != nullptr: inputs are already on the devices.
// (Runs on the worker threads BETWEEN entry points, i.e. without the contexts' locks: failures of the HIP calls made here go
// to the calling thread's message slot only (chk_local); a context handed out by hp_node_ctx must not be switched to another
// stream while a node call is running.)
int sharded_run(hp_node_sharded *p, const uint64_t *h_ct1, const uint64_t *h_ct2, const uint64_t *const *d_ct1,
                const uint64_t *const *d_ct2, uint64_t *const *d_key, uint64_t *h_out, uint64_t *const *d_out) {
    hp_node *node = p->node;
    const size_t W = node->ctx.size(), L = p->L, B = p->batch, logn = p->logn, n = (size_t)1 << logn;
    const uint64_t *mext = p->mext.data();
    const uint64_t inner_t = p->t ? 1 : 0;   // bgv.h:32: relinearize's inner mod switch sees plain_modulus == 1
    const size_t own_p = owner_of(p, L), own_q = owner_of(p, L - 1);
    std::unique_ptr<std::atomic<int>[]> status(new std::atomic<int>[W]);
    for (size_t r = 0; r < W; r++) status[r] = HP_OK;
    auto all_ok = [&] {
        for (size_t r = 0; r < W; r++)
            if (status[r].load()) return false;
        return true;
    };
    const auto everyone = [](size_t) { return true; };
    return run_all(node, [&](size_t r) {
        hp_ctx *c = node->ctx[r];
        auto &R = p->rk[r];
        hipStream_t s = (hipStream_t)hp_ctx_get_stream(c);
        (void)hipSetDevice(node->devices[r]);
        const size_t k0 = p->own[r].first, k1 = p->own[r].second;
        const size_t a0 = std::min(k0, L), a1 = std::min(k1, L);             // owned ciphertext limbs
        const size_t b0 = std::min(k0, L - 1), b1 = std::min(k1, L - 1);     // ... that survive the final drop
        int rc = HP_OK;
        auto step = [&](int code) {
            if (!rc && code) {
                rc = code;
                status[r] = code;
            }
        };
        // a rank that failed keeps meeting the others at the barriers (with nothing enqueued)
        const uint64_t *ct1 = d_ct1 ? d_ct1[r] : R.ct1, *ct2 = d_ct2 ? d_ct2[r] : R.ct2;
        if (!d_ct1) {
            step(hp_memcpy_h2d(c, R.ct1, h_ct1, B * 2 * L * n * 8));
            step(hp_memcpy_h2d(c, R.ct2, h_ct2, B * 2 * L * n * 8));
        }
        const uint64_t *d2 = R.quad + 2 * L * n;   // polynomial 2 of each quadratic ciphertext: stride 3L limbs
        // stage 1: tensor product and strict coefficients of the owned digits              ckks/arith.cpp:55-62, rgsw.cpp:103-105
        if (!rc) step(hp_dev_mult_low_level_range(c, logn, L, mext, B, a0, a1, ct1, ct2, R.quad));
        if (!rc) step(hp_dev_ks_coef_range(c, logn, L, mext, B, a0, a1, d2, 3 * L, R.coef));
        // One exchange in lockstep.  `mine`: this rank's buffer of the exchanged object (its own part is there already; the others'
        // parts arrive there); `buf_of(d)`: rank d's buffer of it; e = 0 / 3: every rank contributes its owned limbs (all-gather
        // shaped), e = 1 / 2: `root` has the whole object (broadcast shaped); receive: this rank needs the others' parts.
        //   HP_TRANSPORT_PEER    the sender writes its part straight into every receiver's buffer (staged through page-locked host
        //                        memory where a pair has no peer access), ordered by the sender's event
        //   HP_TRANSPORT_RCCL    ncclAllGather of the packed, padded parts / ncclBroadcast from the root on the ranks' streams
        //   HP_TRANSPORT_PACKED  the packed parts moved by plain device copies: the collective's data movement without RCCL
        //                        (ranks that share a GPU cannot form a communicator)
        auto exchange = [&](size_t e, bool i_send, uint64_t *mine, const std::function<uint64_t *(size_t)> &buf_of, bool receive, size_t root) {
            const bool gather = e == 0 || e == 3;
            const size_t gi = e == 3 ? 1 : 0, pad = p->pad[gi];
            const Block own = block_of(p, e, r);
            if (p->transport == HP_TRANSPORT_PEER) {
                if (!rc && i_send)
                    step(send_block(p, r, e, s, mine, buf_of, gather && !(e == 3 && !d_out) ? std::function<bool(size_t)>(everyone)
                                                                                             : std::function<bool(size_t)>([&](size_t d) { return gather ? d == 0 : true; })));
                if (!rc) step(chk_local(c, hipEventRecord(R.ev[e], s), "event"));
                node->barrier->wait();
                if (all_ok() && !rc && receive) {
                    if (gather) {
                        for (size_t d = 0; d < W && !rc; d++)
                            if (d != r) step(recv_block(p, r, e, s, d, mine));
                    } else if (r != root) {
                        step(recv_block(p, r, e, s, root, mine));
                    }
                }
                return;
            }
            // packed transports: own limbs [lo, hi) of every row -> pack[rows][pad][N]
            if (!rc && gather && i_send && !own.empty())
                step(chk_local(c, hipMemcpy2DAsync(R.pack[gi], pad * own.n * 8, (const char *)mine + own.lo * own.n * 8, own.pitch(), own.width(), own.rows,
                                                   hipMemcpyDeviceToDevice, s), "pack owned limbs"));
            if (!rc) step(chk_local(c, hipEventRecord(R.ev[e], s), "event"));
            node->barrier->wait();
            if (!all_ok() || rc) return;     // (every rank sees the same verdict here: all of them make the collective call, or none)
            const size_t rows = gather ? (e == 0 ? B : 2 * B) : 1, part = rows * pad * n;   // words of one rank's packed part
            if (p->transport == HP_TRANSPORT_RCCL) {
                Rccl &X = rccl();
                const ncclResult_t er = gather ? X.AllGather(R.pack[gi], R.gath[gi], part, ncclUint64, node->comms[r], s)
                                               : X.Broadcast(mine, mine, 2 * B * n, ncclUint64, (int)root, node->comms[r], s);
                if (er != ncclSuccess) step(fail_local(c, HP_EHIP, std::string(gather ? "ncclAllGather: " : "ncclBroadcast: ") + X.GetErrorString(er)));
            } else if (receive) {
                for (size_t d = 0; d < W && !rc; d++) {
                    if (d == r || (gather ? block_of(p, e, d).empty() : d != root)) continue;
                    step(chk_local(c, hipStreamWaitEvent(s, p->rk[d].ev[e], 0), "wait for a peer's exchange"));
                    if (rc) break;
                    if (gather)
                        step(chk_local(c, hipMemcpyPeerAsync(R.gath[gi] + d * part, node->devices[r], p->rk[d].pack[gi], node->devices[d], part * 8, s),
                                       "copy of a peer's packed limbs"));
                    else
                        step(chk_local(c, hipMemcpyPeerAsync(mine, node->devices[r], buf_of(d), node->devices[d], 2 * B * n * 8, s), "copy of the root's limb"));
                }
            }
            // the others' limbs out of gath[W][rows][pad][N] into their places in `mine`
            if (!rc && gather && receive)
                for (size_t d = 0; d < W && !rc; d++) {
                    const Block b = block_of(p, e, d);
                    if (d == r || b.empty()) continue;
                    step(chk_local(c, hipMemcpy2DAsync((char *)mine + b.lo * b.n * 8, b.pitch(), R.gath[gi] + d * part, pad * b.n * 8, b.width(), b.rows,
                                                       hipMemcpyDeviceToDevice, s), "unpack a peer's limbs"));
                }
        };
        // exchange 1: owned coefficient limbs into every peer's coef buffer (the all-gather of the key-switch digits)
        exchange(0, true, R.coef, [&](size_t d) { return p->rk[d].coef; }, true, 0);
        // stage 2: digits + inner product for the owned output moduli; the owner of p prepares the coefficients of its limb
        if (!rc && all_ok()) step(hp_dev_ks_inner_range_strict(c, logn, L, mext, B, k0, k1, R.coef, d2, 3 * L, d_key[r], R.ks));
        if (!rc && all_ok() && r == own_p) step(hp_dev_drop_coeffs(c, logn, L + 1, mext, inner_t, 2 * B, R.ks, R.c_p));
        exchange(1, r == own_p && all_ok(), R.c_p, [&](size_t d) { return p->rk[d].c_p; }, true, own_p);
        // stage 3: drop p on the owned limbs (+= d0, d1); the owner of q_{L-1} prepares that limb's coefficients
        if (!rc && all_ok())
            step(hp_dev_drop_apply_range_strict(c, logn, L + 1, mext, inner_t, 2 * B, a0, a1, R.ks, R.c_p, R.quad, L, 3 * L, 3, R.relin));
        if (!rc && all_ok() && r == own_q) step(hp_dev_drop_coeffs(c, logn, L, mext, p->t, 2 * B, R.relin, R.c_q));
        exchange(2, r == own_q && all_ok(), R.c_q, [&](size_t d) { return p->rk[d].c_q; }, true, own_q);
        // stage 4: drop q_{L-1} on the owned limbs; result limbs go to rank 0 (and to the caller's per-rank buffers)
        uint64_t *out = d_out ? d_out[r] : R.out;
        if (!rc && all_ok()) step(hp_dev_drop_apply_range_strict(c, logn, L, mext, p->t, 2 * B, b0, b1, R.relin, R.c_q, nullptr, 0, 0, 0, out));
        exchange(3, all_ok() && (d_out || r != 0 || p->transport != HP_TRANSPORT_PEER), out,
                 [&](size_t d) { return d_out ? d_out[d] : p->rk[d].out; }, d_out || r == 0, 0);
        if (!rc && all_ok() && !d_out && r == 0) step(hp_memcpy_d2h(c, h_out, R.out, B * 2 * (L - 1) * n * 8));
        // the call returns with every stream drained: the next call may overwrite peers' buffers at once
        step(hp_sync(c));
        node->barrier->wait();
        if (!rc && !all_ok()) rc = HP_PEER_FAILED;   // another rank failed: this rank's result is not valid either
        return rc;
    });
}

} // namespace

extern "C" {

void hp_node_sharded_destroy(hp_node_sharded *plan) { free_plan(plan); }

int hp_node_sharded_create(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus, size_t batch,
                           hp_node_sharded **out) {
    if (!node || !moduli_ext || !out) return HP_EINVAL;
    if (logn < 1 || logn > 16 || L < 2 || L + 1 > HP_MAX_LIMBS || batch == 0) return node_fail(node, HP_EINVAL, "invalid shape");
    std::lock_guard<std::mutex> lk(node->mu);
    hp_node_sharded *p = new (std::nothrow) hp_node_sharded();
    if (!p) return HP_ENOMEM;
    p->node = node; p->logn = logn; p->L = L; p->batch = batch; p->t = plain_modulus;
    p->mext.assign(moduli_ext, moduli_ext + L + 1);
    const size_t W = node->ctx.size(), n = (size_t)1 << logn;
    // contiguous ownership ranges of the L+1 extended moduli, sizes differing by at most one, the LARGER ranges first: the
    // special prime (the most expensive output modulus: L digit transforms per polynomial instead of L-1) is the last
    // modulus and therefore sits in a smallest range.  With more ranks than moduli the tail ranks own nothing.
    p->own.resize(W);
    for (size_t r = 0; r < W; r++) slice_of(L + 1, W, r, &p->own[r].first, &p->own[r].second);
    p->rk.resize(W);
    p->transport = node->transport;
    if (p->transport != HP_TRANSPORT_PEER)
        for (size_t r = 0; r < W; r++) {
            const Block b0 = block_of(p, 0, r), b3 = block_of(p, 3, r);
            p->pad[0] = std::max(p->pad[0], b0.empty() ? (size_t)0 : b0.hi - b0.lo);
            p->pad[1] = std::max(p->pad[1], b3.empty() ? (size_t)0 : b3.hi - b3.lo);
        }
    int rc = run_all(node, [&](size_t r) {
        hp_ctx *c = node->ctx[r];
        auto &R = p->rk[r];
        const size_t pk0 = batch * p->pad[0] * n, pk3 = 2 * batch * p->pad[1] * n;   // (zero words: never allocated)
        struct { uint64_t **ptr; size_t words; } bufs[] = {
            {&R.ct1, batch * 2 * L * n}, {&R.ct2, batch * 2 * L * n}, {&R.quad, batch * 3 * L * n}, {&R.coef, batch * L * n},
            {&R.ks, batch * 2 * (L + 1) * n}, {&R.c_p, 2 * batch * n}, {&R.relin, batch * 2 * L * n}, {&R.c_q, 2 * batch * n},
            {&R.out, batch * 2 * (L - 1) * n}, {&R.pack[0], pk0}, {&R.gath[0], W * pk0}, {&R.pack[1], pk3}, {&R.gath[1], W * pk3}};
        for (auto &b : bufs) {
            if (b.words == 0) continue;
            int rc2 = hp_dev_alloc(c, b.words * 8, (void **)b.ptr);
            if (rc2) return rc2;
        }
        (void)hipSetDevice(node->devices[r]);
        for (auto &e : R.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return (int)HP_EHIP;
        // page-locked staging for the blocks this rank sends, only if some peer cannot be written directly
        bool all_direct = true;
        for (size_t d = 0; d < W; d++) all_direct = all_direct && direct(node, r, d);
        if (!all_direct)
            for (size_t e = 0; e < 4; e++) {
                const Block b = block_of(p, e, r);
                if (b.empty()) continue;
                int rc2 = hp_host_alloc(c, b.bytes(), (void **)&R.hstage[e]);
                if (rc2) return rc2;
            }
        return (int)HP_OK;
    });
    if (rc) {
        free_plan(p);
        return rc;
    }
    *out = p;
    return HP_OK;
}

int hp_node_sharded_range(const hp_node_sharded *plan, size_t rank, size_t *k0, size_t *k1) {
    if (!plan || !k0 || !k1 || rank >= plan->own.size()) return HP_EINVAL;
    *k0 = plan->own[rank].first;
    *k1 = plan->own[rank].second;
    return HP_OK;
}

int hp_node_sharded_mult(hp_node_sharded *plan, const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key, uint64_t *h_out) {
    if (!plan || !h_ct1 || !h_ct2 || !d_key || !h_out) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(plan->node->mu);
    return sharded_run(plan, h_ct1, h_ct2, nullptr, nullptr, d_key, h_out, nullptr);
}

int hp_node_sharded_mult_dev(hp_node_sharded *plan, const uint64_t *const *d_ct1, const uint64_t *const *d_ct2, uint64_t *const *d_key,
                             uint64_t *const *d_out) {
    if (!plan || !d_ct1 || !d_ct2 || !d_key || !d_out) return HP_EINVAL;
    std::lock_guard<std::mutex> lk(plan->node->mu);
    return sharded_run(plan, nullptr, nullptr, d_ct1, d_ct2, d_key, nullptr, d_out);
}

} // extern "C"
