// batched_forms.cpp -- hehub_amd_ext.hpp: the batched forms of the scheme-level calls (std::vector<Ct> in, one engine call per device
// rank, std::vector<Ct> out); element i is word for word what hehub's single call returns for element i.
#include "layer.hpp"

namespace hehub {

using namespace detail;

// =====================================================================================================
// hehub_amd_ext.hpp: batched forms of the scheme-level calls
// =====================================================================================================
namespace amd {

namespace {

// the shape all members of a batch share (both halves of every ciphertext: dimension, limbs, moduli), or false
template <class Ct> bool uniform_shape(const std::vector<Ct> &cts, size_t &n, size_t &L, std::vector<u64> &q) {
    if (cts.empty()) return false;
    n = cts[0][0].dimension();
    L = cts[0][0].component_count();
    q = cts[0][0].modulus_vec();
    q.resize(L);
    if (L == 0 || n < 2) return false;
    for (const Ct &ct : cts)
        for (int h = 0; h < 2; h++) {
            if (ct[h].dimension() != n || ct[h].component_count() != L) return false;
            std::vector<u64> m(ct[h].modulus_vec());
            m.resize(L);
            if (m != q) return false;
        }
    return true;
}
template <class Ct> std::vector<const RnsIntVec *> halves(const std::vector<Ct> &cts) {
    std::vector<const RnsIntVec *> v;
    v.reserve(2 * cts.size());
    for (const Ct &ct : cts) { v.push_back(&ct[0]); v.push_back(&ct[1]); }
    return v;
}
// B result ciphertexts of the given shape, no words yet
template <class Ct> std::vector<Ct> result_shells(size_t B, size_t n, size_t L, const std::vector<u64> &q) {
    std::vector<Ct> out;
    out.reserve(B);
    for (size_t i = 0; i < B; i++)
        out.emplace_back(RlweCt{result_poly(n, L, q, PolyRepForm::value), result_poly(n, L, q, PolyRepForm::value)});
    return out;
}
// the elements of a batch that one device rank runs (ascending)
typedef std::vector<size_t> Part;
// ... the elements of `part` become views of u64[part.size()][2][L][N], the block one rank's engine call filled
template <class Ct> void bind_part(std::vector<Ct> &out, const Part &part, const Dst &d, size_t L) {
    std::vector<RnsIntVec *> polys;
    for (size_t i : part) { polys.push_back(&out[i][0]); polys.push_back(&out[i][1]); }
    Access::bind_many(polys, d, L);
}
template <class Ct> std::vector<const RnsIntVec *> halves(const std::vector<Ct> &cts, const Part &part) {
    std::vector<const RnsIntVec *> v;
    v.reserve(2 * part.size());
    for (size_t i : part) { v.push_back(&cts[i][0]); v.push_back(&cts[i][1]); }
    return v;
}
// the device rank element i's ciphertext lives on (-1: only on the host so far)
template <class Ct> int home_of(const Ct &ct) {
    for (int h = 0; h < 2; h++)
        if (const BlockRef *r = Access::home(ct[h]))
            if (*r) return home_rank(**r);
    return -1;
}
// A batch is cut into parts, one per device rank (no collective; with one rank: the whole batch), and f(part) runs for each
// inside a scope on lane 0 of its rank: a batch fills a GPU by itself, and only one lane per rank grows a batch-sized workspace.  An
// element whose operand already lives on a rank stays THERE (nothing crosses between ranks: a batched call after a loop of single calls,
// or on the results of an earlier batched call, finds its operands where they are); the elements that live only on the host fill the
// ranks up to an even share, contiguously (a batch of host objects: batch / ranks each, the contiguous slices of SURVEY.md 8e).  The
// parts' engine calls are enqueued one after the other and run side by side on their devices.
template <class Home, class F> void for_parts(size_t B, Home &&home, F &&f) {
    (void)engine();
    const size_t nd = (size_t)lane_set().ndev;
    std::vector<Part> part(nd);
    Part homeless;
    for (size_t i = 0; i < B; i++) {
        const int h = nd == 1 ? 0 : home(i);
        if (h >= 0 && (size_t)h < nd) part[(size_t)h].push_back(i);
        else homeless.push_back(i);
    }
    const size_t share = (B + nd - 1) / nd;
    size_t r = 0;
    for (size_t i : homeless) {
        while (r + 1 < nd && part[r].size() >= share) r++;
        part[r].push_back(i);
    }
    for (size_t k = 0; k < nd; k++) {
        if (part[k].empty()) continue;
        std::sort(part[k].begin(), part[k].end());
        OpScope op({}, 0, (int)k);
        f(part[k]);
    }
}
void same_size(size_t a, size_t b) {
    if (a != b) throw std::invalid_argument("hehub_amd: the two batches have different sizes.");
}

// the checks of mult_low_level (ckks/arith.cpp:55-62 -> operator*, rns.h:253-270), member by member; false: not one common shape
template <class Ct> bool mult_args_ok(const std::vector<Ct> &a, const std::vector<Ct> &b, size_t &n, size_t &L, std::vector<u64> &q) {
    size_t nb, Lb;
    std::vector<u64> qb;
    const bool ua = uniform_shape(a, n, L, q), ub = uniform_shape(b, nb, Lb, qb);
    if (!ua || !ub || nb != n || Lb != L) return false;
    for (size_t i = 0; i < a.size(); i++)
        for (int h = 0; h < 2; h++) {
            if (a[i][h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand a is in coefficient form.");
            if (b[i][h].rep_form == PolyRepForm::coeff) throw std::invalid_argument("Operand b is in coefficient form.");
        }
    if (q != qb) throw std::invalid_argument("Operands' moduli mismatch.");
    return true;
}

// ckks::mult / bgv mult_low_level + relinearize [+ drop of q_last] on a batch; bgv: t = the common plain modulus
template <class Ct>
std::vector<Ct> mult_batch(const std::vector<Ct> &a, const std::vector<Ct> &b, const RlweKsk &key, bool drop, bool bgv, u64 t, size_t n, size_t L,
                           const std::vector<u64> &q) {
    const size_t B = a.size();
    size_t logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(result_poly(n, L, q, PolyRepForm::value), key, mext);
    if (bgv && L0 != L) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    if (drop && L == 1) throw std::invalid_argument("Unable to drop the only one prime.");
    const size_t Lout = drop ? L - 1 : L;
    std::vector<Ct> out = result_shells<Ct>(B, n, Lout, q);
    for_parts(B, [&](size_t i) { const int h = home_of(a[i]); return h >= 0 ? h : home_of(b[i]); }, [&](const Part &part) {
        const size_t S = part.size();
        DevKey dk(key, L0, n);   // (the key on this part's device: assembled there on first use, cached per rank)
        Src d1 = Access::batch_in(halves(a, part), L), d2 = Access::batch_in(halves(b, part), L);
        Dst dout(S * 2 * Lout * n);
        if (drop) {
            if (bgv) check(hp_dev_bgv_mult_relin_modswitch(cur(), logn, L, mext.data(), t, S, d1.p, d2.p, dk.p(), dout.p));
            else check(hp_dev_ckks_mult_relin_rescale_at(cur(), logn, L, L0, mext.data(), S, d1.p, d2.p, dk.p(), dout.p));
        } else {
            Dst dq(S * 3 * L * n);
            check(hp_dev_mult_low_level(cur(), logn, L, q.data(), S, d1.p, d2.p, dq.p));
            if (bgv) check(hp_dev_bgv_relinearize(cur(), logn, L, mext.data(), 1 /* bgv.h:32 */, S, dq.p, dk.p(), dout.p));
            else check(hp_dev_ckks_relinearize_at(cur(), logn, L, L0, mext.data(), S, dq.p, dk.p(), dout.p));
        }
        bind_part(out, part, dout, Lout);
    });
    return out;
}

// rescale_inplace / mod_switch_inplace by one prime on a batch of one shape
template <class Ct> void drop_batch(std::vector<Ct> &cts, bool bgv, u64 t, size_t n, size_t L, const std::vector<u64> &q) {
    const size_t B = cts.size();
    size_t logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    for_parts(B, [&](size_t i) { return home_of(cts[i]); }, [&](const Part &part) {
        const size_t S = part.size();
        Src din = Access::batch_in(halves(cts, part), L);
        Dst dout(S * 2 * (L - 1) * n);
        if (bgv) check(hp_dev_bgv_mod_switch(cur(), logn, L, q.data(), t, S, din.p, dout.p));
        else check(hp_dev_ckks_rescale(cur(), logn, L, q.data(), S, din.p, dout.p));
#ifdef HEHUB_AMD_BIND_REFERENCE
        limb_copies_wait();   // (remove_components hands limb blocks back to hehub's pool: their uploads must have happened, see drop_last_prime)
#endif
        std::vector<RnsIntVec *> polys;
        for (size_t i : part)
            for (int h = 0; h < 2; h++) {
                cts[i][h].remove_components();
                polys.push_back(&cts[i][h]);
            }
        Access::bind_many(polys, dout, L - 1);
    });
}

std::vector<ckks::CkksCt> ckks_mult(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key, bool rescale) {
    same_size(a.size(), b.size());
    size_t n, L;
    std::vector<u64> q;
    std::vector<ckks::CkksCt> out;
    if (a.empty()) return out;
    if (!mult_args_ok(a, b, n, L, q)) {   // no common shape: the loop of single calls, with their checks
        for (size_t i = 0; i < a.size(); i++) {
            out.push_back(ckks::mult(a[i], b[i], key));
            if (rescale) ckks::rescale_inplace(out.back());
        }
        return out;
    }
    out = mult_batch(a, b, key, rescale, false, 0, n, L, q);
    for (size_t i = 0; i < a.size(); i++) {
        out[i].scaling_factor = a[i].scaling_factor * b[i].scaling_factor;   // ckks/arith.cpp:61, :68
        if (rescale) out[i].scaling_factor /= q[L - 1];                      // rescaling.cpp:87
    }
    return out;
}

std::vector<bgv::BgvCt> bgv_mult(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key, bool drop) {
    same_size(a.size(), b.size());
    size_t n, L;
    std::vector<u64> q;
    std::vector<bgv::BgvCt> out;
    if (a.empty()) return out;
    bool one_t = true;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].plain_modulus != b[i].plain_modulus) throw std::invalid_argument("Plain moduli mismatch.");   // bgv/arith.cpp:60-62
        one_t = one_t && a[i].plain_modulus == a[0].plain_modulus;
    }
    if (!one_t || !mult_args_ok(a, b, n, L, q)) {
        for (size_t i = 0; i < a.size(); i++) {
            out.push_back(bgv::relinearize(bgv::mult_low_level(a[i], b[i]), key));
            if (drop) bgv::mod_switch_inplace(out.back());
        }
        return out;
    }
    out = mult_batch(a, b, key, drop, true, a[0].plain_modulus, n, L, q);
    for (auto &ct : out) ct.plain_modulus = a[0].plain_modulus;
    return out;
}

std::vector<ckks::CkksCt> ckks_key_switched(const std::vector<ckks::CkksCt> &cts, const RlweKsk &key, bool conj, size_t step) {
    size_t n, L;
    std::vector<u64> q;
    std::vector<ckks::CkksCt> out;
    if (cts.empty()) return out;
    if (!uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) out.push_back(conj ? ckks::conjugate(ct, key) : ckks::rotate(ct, key, step));
        return out;
    }
    for (auto &ct : cts)
        for (int h = 0; h < 2; h++)
            if (ct[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext;
    const size_t L0 = check_ext_prod(cts[0][1], key, mext);
    const size_t logn = cts[0][1].log_dimension(), B = cts.size();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    for_parts(B, [&](size_t i) { return home_of(cts[i]); }, [&](const Part &part) {
        DevKey dk(key, L0, n);
        Src din = Access::batch_in(halves(cts, part), L);
        Dst dout(part.size() * 2 * L * n);
        if (conj) check(hp_dev_ckks_conjugate_at(cur(), logn, L, L0, mext.data(), part.size(), din.p, dk.p(), dout.p));
        else check(hp_dev_ckks_rotate_at(cur(), logn, L, L0, mext.data(), part.size(), step, din.p, dk.p(), dout.p));
        bind_part(out, part, dout, L);
    });
    for (size_t i = 0; i < B; i++) out[i].scaling_factor = cts[i].scaling_factor;
    return out;
}

// ckks::rotate(cts[i], *keys[i], steps[i]) for every i as ONE engine call (hp_dev_ckks_rotate_many); a batch that is not of one shape,
// or whose keys were not all made for the same number of moduli, runs as the loop of single calls
std::vector<ckks::CkksCt> ckks_rotate_many(const std::vector<const ckks::CkksCt *> &cts, const std::vector<const RlweKsk *> &keys,
                                           const std::vector<size_t> &steps) {
    same_size(cts.size(), keys.size());
    same_size(cts.size(), steps.size());
    std::vector<ckks::CkksCt> out;
    if (cts.empty()) return out;
    for (const RlweKsk *k : keys)
        if (!k) throw std::invalid_argument("Empty RGSW ciphertext.");
    const size_t n = (*cts[0])[0].dimension(), L = (*cts[0])[0].component_count(), B = cts.size();
    std::vector<u64> q((*cts[0])[0].modulus_vec());
    q.resize(L);
    bool uniform = L > 0 && n >= 2;
    for (size_t i = 0; uniform && i < B; i++) {
        for (int h = 0; h < 2 && uniform; h++) {
            const RnsPolynomial &p = (*cts[i])[h];
            std::vector<u64> m(p.modulus_vec());
            m.resize(L);
            uniform = p.dimension() == n && p.component_count() == L && m == q;
        }
        uniform = uniform && keys[i]->size() == keys[0]->size();
    }
    if (!uniform) {
        for (size_t i = 0; i < B; i++) out.push_back(ckks::rotate(*cts[i], *keys[i], steps[i]));
        return out;
    }
    for (const ckks::CkksCt *ct : cts)
        for (int h = 0; h < 2; h++)
            if ((*ct)[h].rep_form != PolyRepForm::value) throw std::invalid_argument("poly_ntt is expected to be in NTT value form");
    std::vector<u64> mext, mext_i;
    const size_t L0 = check_ext_prod((*cts[0])[1], *keys[0], mext);
    for (size_t i = 1; i < B; i++)
        if (check_ext_prod((*cts[i])[1], *keys[i], mext_i) != L0 || mext_i != mext) throw std::invalid_argument("Inconsistent RGSW ciphertext.");
    const size_t logn = (*cts[0])[1].log_dimension();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    // ONE ciphertext under many keys (src/circuits/linear_algebra.h:123-130) is one operand: it stays on its device, the batch is not cut
    // (a part elsewhere would drag the vector and every key of the part over).  Different ciphertexts: a part per rank (for_parts).
    bool one_ct = true;
    for (size_t i = 1; i < B; i++) one_ct = one_ct && cts[i] == cts[0];
    auto run = [&](const Part &part) {
        std::vector<DevKey> dks;
        dks.reserve(part.size());
        std::vector<const u64 *> kp;
        std::vector<size_t> st;
        for (size_t a = 0; a < part.size(); a++) {   // (a key that appears several times is assembled once: the key cache, or the earlier element)
            const size_t i = part[a];
            st.push_back(steps[i]);
            size_t same = a;
            for (size_t b = 0; b < a && same == a; b++)
                if (keys[part[b]] == keys[i]) same = b;
            if (same < a) { kp.push_back(kp[same]); continue; }
            dks.emplace_back(*keys[i], L0, n);
            kp.push_back(dks.back().p());
        }
        std::vector<const u64 *> polys;   // the operands are read where they are: the same object may appear many times
        std::vector<Src> holds;
        for (size_t i : part)
            for (int h = 0; h < 2; h++) {
                holds.push_back(Access::in((*cts[i])[h], L));
                polys.push_back(holds.back().p);
            }
        Dst dout(part.size() * 2 * L * n);
        check(hp_dev_ckks_rotate_many_rows(cur(), logn, L, L0, mext.data(), part.size(), st.data(), nullptr, polys.data(), kp.data(), dout.p));
        bind_part(out, part, dout, L);
    };
    if (one_ct) {
        Part all(B);
        for (size_t i = 0; i < B; i++) all[i] = i;
        OpScope op({Access::home((*cts[0])[0]), Access::home((*cts[0])[1])}, 0);
        run(all);
    } else {
        for_parts(B, [&](size_t i) { return home_of(*cts[i]); }, run);
    }
    for (size_t i = 0; i < B; i++) out[i].scaling_factor = cts[i]->scaling_factor;
    return out;
}

std::vector<ckks::CkksCt> ckks_addsub(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, bool sub) {
    same_size(a.size(), b.size());
    size_t n, L, nb, Lb;
    std::vector<u64> q, qb;
    std::vector<ckks::CkksCt> out;
    if (a.empty()) return out;
    bool fast = uniform_shape(a, n, L, q) && uniform_shape(b, nb, Lb, qb) && nb == n && Lb >= L;   // (b may carry more limbs: rns.cpp:59-72)
    if (fast) {
        qb.resize(L);
        fast = qb == q;
    }
    for (size_t i = 0; i < a.size() && fast; i++)
        for (int h = 0; h < 2; h++) fast = a[i][h].rep_form == b[i][h].rep_form && a[i][h].rep_form == a[0][0].rep_form;
    for (size_t i = 0; i < a.size() && fast; i++) fast = std::abs(a[i].scaling_factor - b[i].scaling_factor) <= std::pow(2.0, -50);
    if (!fast) {   // the loop of single calls throws what hehub throws, where hehub throws it
        for (size_t i = 0; i < a.size(); i++) out.push_back(sub ? ckks::sub(a[i], b[i]) : ckks::add(a[i], b[i]));
        return out;
    }
    const size_t B = a.size();
    out = result_shells<ckks::CkksCt>(B, n, L, q);
    for_parts(B, [&](size_t i) { const int h = home_of(a[i]); return h >= 0 ? h : home_of(b[i]); }, [&](const Part &part) {
        Src da = Access::batch_in(halves(a, part), L), db = Access::batch_in(halves(b, part), L);
        Dst dout(part.size() * 2 * L * n);
        dev_binary(sub ? Bin::sub : Bin::add, n, L, q.data(), 2 * part.size(), da.p, db.p, dout.p);
        bind_part(out, part, dout, L);
    });
    for (size_t i = 0; i < B; i++) {
        out[i].scaling_factor = a[i].scaling_factor;
        for (int h = 0; h < 2; h++) out[i][h].rep_form = a[i][h].rep_form;
    }
    return out;
}

} // namespace

std::vector<ckks::CkksCt> mult(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key) { return ckks_mult(a, b, key, false); }
std::vector<ckks::CkksCt> mult_rescale(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &key) { return ckks_mult(a, b, key, true); }
std::vector<bgv::BgvCt> mult(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key) { return bgv_mult(a, b, key, false); }
std::vector<bgv::BgvCt> mult_mod_switch(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &key) { return bgv_mult(a, b, key, true); }
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &rot_key, size_t step) { return ckks_key_switched(cts, rot_key, false, step); }
std::vector<ckks::CkksCt> conjugate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &conj_key) { return ckks_key_switched(cts, conj_key, true, 0); }
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const std::vector<const RlweKsk *> &rot_keys, const std::vector<size_t> &steps) {
    std::vector<const ckks::CkksCt *> p;
    for (auto &ct : cts) p.push_back(&ct);
    return ckks_rotate_many(p, rot_keys, steps);
}
std::vector<ckks::CkksCt> rotate(const ckks::CkksCt &ct, const std::vector<const RlweKsk *> &rot_keys, const std::vector<size_t> &steps) {
    return ckks_rotate_many(std::vector<const ckks::CkksCt *>(rot_keys.size(), &ct), rot_keys, steps);
}
std::vector<ckks::CkksCt> add(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b) { return ckks_addsub(a, b, false); }
std::vector<ckks::CkksCt> sub(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b) { return ckks_addsub(a, b, true); }

void rescale_inplace(std::vector<ckks::CkksCt> &cts) {
    size_t n, L;
    std::vector<u64> q;
    if (cts.empty()) return;
    for (auto &ct : cts) check_ct_wellformed(ct);   // rescaling.cpp:15-29
    if (!uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) ckks::rescale_inplace(ct);
        return;
    }
    drop_batch(cts, false, 0, n, L, q);
    for (auto &ct : cts) ct.scaling_factor /= q[L - 1];
}

void mod_switch_inplace(std::vector<bgv::BgvCt> &cts) {
    size_t n, L;
    std::vector<u64> q;
    if (cts.empty()) return;
    for (auto &ct : cts) check_ct_wellformed(ct);   // mod_switch.cpp:14-28
    bool one_t = true;
    for (auto &ct : cts) one_t = one_t && ct.plain_modulus == cts[0].plain_modulus;
    if (!one_t || !uniform_shape(cts, n, L, q)) {
        for (auto &ct : cts) bgv::mod_switch_inplace(ct);
        return;
    }
    drop_batch(cts, true, cts[0].plain_modulus, n, L, q);
}

} // namespace amd

} // namespace hehub

