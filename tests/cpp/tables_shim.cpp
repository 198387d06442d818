// tables_shim.cpp -- C entry points around hehub_amd/csrc/hp_tables.cpp so the CPU test-suite can check the
// host-side table builders (values AND kernel-order layouts) without a GPU.  Test infrastructure only.
#include "../../hehub_amd/csrc/hp_tables.h"

#include <cstring>

extern "C" {

int tbl_check(uint64_t q, size_t logn) { return hp::check_ntt_modulus(q, logn).empty() ? 0 : 1; }
uint64_t tbl_root(uint64_t q, size_t logn) { return hp::unity_root_2n(q, logn); }
uint64_t tbl_inverse(uint64_t e, uint64_t p) { return hp::inverse_mod_prime(e, p); }
void tbl_consts(uint64_t q, uint64_t *out9) {
    hp::ModConsts c = hp::make_consts(q);
    uint64_t v[9] = {c.q, c.two_q, c.neg_q, c.mqinv, c.r64, c.r64h, c.barrett_c, c.k, c.fix};
    std::memcpy(out9, v, sizeof(v));
}
// out sizes (in pairs = 2 u64): fwd_ref N, inv_ref 2N, fwd_fast 31*2^a + 31*N/32, inv_fast 31 + 31*32 + 31*N/32
void tbl_fwd_ref(uint64_t q, size_t logn, uint64_t *out) {
    std::vector<hp::Pair> t; hp::build_fwd_ref(q, logn, t); std::memcpy(out, t.data(), t.size() * sizeof(hp::Pair));
}
void tbl_inv_ref(uint64_t q, size_t logn, uint64_t *out) {
    std::vector<hp::Pair> t; hp::build_inv_ref(q, logn, t); std::memcpy(out, t.data(), t.size() * sizeof(hp::Pair));
}
void tbl_fwd_fast(uint64_t q, size_t logn, uint64_t *out) {
    std::vector<hp::Pair> r, t; hp::build_fwd_ref(q, logn, r); hp::build_fwd_fast(r, logn, t);
    std::memcpy(out, t.data(), t.size() * sizeof(hp::Pair));
}
void tbl_inv_fast(uint64_t q, size_t logn, uint64_t *out) {
    std::vector<hp::Pair> r, t; hp::build_inv_ref(q, logn, r); hp::build_inv_fast(r, logn, t);
    std::memcpy(out, t.data(), t.size() * sizeof(hp::Pair));
}
}
