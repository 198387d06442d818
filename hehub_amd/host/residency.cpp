// residency.cpp -- own-mirror build: the staging of a vector's limbs, the RnsIntVec members (rns.h:15-115 mirrored in hehub.hpp); both
// builds: the gather of sibling views into one block.
#include "layer.hpp"

namespace hehub {

namespace amd {

#ifndef HEHUB_AMD_BIND_REFERENCE
unsigned long long next_stamp() {
    static unsigned long long s = 0;
    return ++s;
}

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


} // namespace amd

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

} // namespace hehub
