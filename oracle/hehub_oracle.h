/*
 * hehub_oracle.h -- CPU ORACLE for the RNS ring-arithmetic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the arithmetic
 * that primihub/hehub performs on the path named by BASELINE.json:north_star.
 * It exists to CHECK the HIP engine (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under hehub_amd/ may include, link,
 * import or execute it.
 *
 * Parity status: PINNED.  Every function here is compared word-for-word with
 * the unmodified reference compiled from /root/reference (oracle/Makefile ->
 * oracle/_ref/libhehub_ref.so) by tests/test_oracle_vs_reference.py, and with
 * the committed golden vectors under tests/golden/ (made by
 * tests/golden/make_golden.py from that same reference build) plus the known
 * answers quoted in SURVEY.md section 8a.
 *
 * Data layouts (all row-major, little-endian u64):
 *   limb              u64[N]
 *   polynomial        u64[L][N]
 *   ciphertext        u64[2][L][N]      (quadratic: u64[3][L][N])
 *   key-switch key    u64[Ld][2][Le][N] with Ld = digits = #ct limbs,
 *                                          Le = Ld + 1 (last = special prime)
 * Citations "file:line" are relative to /root/reference.
 */
#ifndef HEHUB_ORACLE_H
#define HEHUB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t orc_u64;

/* ---- scalar helpers ------------------------------------------------- */
orc_u64 orc_harvey_quotient(orc_u64 b, orc_u64 q);           /* floor(b*2^64/q) */
orc_u64 orc_mul_mod_harvey_lazy(orc_u64 q, orc_u64 a, orc_u64 b, orc_u64 b_harvey);
orc_u64 orc_inverse_mod_prime(orc_u64 elem, orc_u64 prime);
orc_u64 orc_pow_mod(orc_u64 q, orc_u64 base, orc_u64 index);
/* returns 0 and writes *root on success, -1 if 2N does not divide q-1 */
int orc_get_2nth_unity_root(orc_u64 q, orc_u64 n, orc_u64 *root);
orc_u64 orc_bit_rev(orc_u64 x, int bit_len);
orc_u64 orc_minus_q_inv_mod_2to64(orc_u64 q);
orc_u64 orc_2to64_mod_q(orc_u64 q);

/* ---- batched modular kernels (mod_arith.h / mod_arith.cpp) ---------- */
void orc_batched_barrett_lazy(orc_u64 q, size_t n, orc_u64 *v);
void orc_batched_barrett(orc_u64 q, size_t n, orc_u64 *v);
void orc_batched_reduce_strict(orc_u64 q, size_t n, orc_u64 *v);
void orc_batched_mul_mod_hybrid_lazy(orc_u64 q, size_t n, const orc_u64 *a,
                                     const orc_u64 *b, orc_u64 *out);
void orc_batched_mul_mod_barrett_lazy(orc_u64 q, size_t n, const orc_u64 *a,
                                      const orc_u64 *b, orc_u64 *out);
/* in128 is n pairs {lo,hi} */
void orc_batched_montgomery_128_lazy(orc_u64 q, size_t n, const orc_u64 *in128,
                                     orc_u64 *out);

/* ---- twiddle tables (ntt.cpp:41-105) -------------------------------- */
/* forward: seq[N], seq_harvey[N].  returns 0, -1 (2N !| q-1), -2 (q > 59 bit) */
int orc_ntt_factors(orc_u64 q, size_t logn, orc_u64 *seq, orc_u64 *seq_harvey);
/* inverse: seq[2N], seq_harvey[2N] */
int orc_intt_factors(orc_u64 q, size_t logn, orc_u64 *seq, orc_u64 *seq_harvey);

/* ---- transforms (ntt.cpp:145-223), in place, one limb --------------- */
int orc_ntt_negacyclic_inplace_lazy(size_t logn, orc_u64 q, orc_u64 *x);
int orc_intt_negacyclic_inplace_lazy(size_t logn, orc_u64 q, orc_u64 *x);
/* drop every cached table (tests that count memory / timing) */
void orc_clear_cache(void);

/* ---- RnsIntVec operators on u64[L][N] (rns.cpp:58-171) -------------- */
void orc_poly_add_inplace(size_t n, size_t L, const orc_u64 *moduli,
                          orc_u64 *self, const orc_u64 *b);
void orc_poly_sub_inplace(size_t n, size_t L, const orc_u64 *moduli,
                          orc_u64 *self, const orc_u64 *b);
void orc_poly_mul(size_t n, size_t L, const orc_u64 *moduli, const orc_u64 *a,
                  const orc_u64 *b, orc_u64 *out);
void orc_poly_scalar_mul_inplace(size_t n, size_t L, const orc_u64 *moduli,
                                 orc_u64 *self, orc_u64 small_scalar);
void orc_poly_rns_scalar_mul_inplace(size_t n, size_t L, const orc_u64 *moduli,
                                     orc_u64 *self, const orc_u64 *rns_scalar);
int orc_poly_ntt(size_t logn, size_t L, const orc_u64 *moduli, orc_u64 *x);
int orc_poly_intt(size_t logn, size_t L, const orc_u64 *moduli, orc_u64 *x);
void orc_poly_reduce_strict(size_t n, size_t L, const orc_u64 *moduli, orc_u64 *x);

/* ---- automorphisms on NTT-form limbs (permutation.cpp:28-75) -------- */
void orc_poly_involution(size_t logn, size_t L, const orc_u64 *in, orc_u64 *out);
void orc_poly_cycle(size_t logn, size_t L, size_t step, const orc_u64 *in,
                    orc_u64 *out);

/* ---- key switch (rgsw.cpp:57-156) ----------------------------------- */
/* moduli_ext has L+1 entries (q_0..q_{L-1}, p); pt u64[L][N] in NTT form;
 * key u64[L][2][L+1][N]; out u64[2][L+1][N]. */
int orc_ext_prod_montgomery(size_t logn, size_t L, const orc_u64 *moduli_ext,
                            const orc_u64 *pt, const orc_u64 *key, orc_u64 *out);

/* ---- drop the last prime of one ciphertext -------------------------- */
/* ct u64[2][L][N] -> out u64[2][L-1][N]   (rescaling.cpp:14-78) */
int orc_ckks_rescale_by_one_prime(size_t logn, size_t L, const orc_u64 *moduli,
                                  const orc_u64 *ct, orc_u64 *out);
/* (mod_switch.cpp:13-78) */
int orc_bgv_mod_drop_one_prime(size_t logn, size_t L, const orc_u64 *moduli,
                               orc_u64 plain_modulus, const orc_u64 *ct,
                               orc_u64 *out);

/* ---- scheme level ---------------------------------------------------- */
/* ct1, ct2 u64[2][L][N] -> out u64[3][L][N] (ckks/arith.cpp:55-62, bgv/arith.cpp:59-69) */
void orc_mult_low_level(size_t n, size_t L, const orc_u64 *moduli,
                        const orc_u64 *ct1, const orc_u64 *ct2, orc_u64 *out);
/* quad u64[3][L][N], key as above -> out u64[2][L][N] (ckks/arith.cpp:64-73) */
int orc_ckks_relinearize(size_t logn, size_t L, const orc_u64 *moduli_ext,
                         const orc_u64 *quad, const orc_u64 *key, orc_u64 *out);
/* bgv/arith.cpp:71-79.  inner_plain_modulus is the plain modulus seen by the
 * internal mod switch; the reference always uses 1 (bgv.h:32) -- pass 1 for
 * reference parity. */
int orc_bgv_relinearize(size_t logn, size_t L, const orc_u64 *moduli_ext,
                        orc_u64 inner_plain_modulus, const orc_u64 *quad,
                        const orc_u64 *key, orc_u64 *out);
/* ckks/arith.cpp:85-93 rotate(ct, rot_key, step) and :75-83 conjugate(ct, conj_key): ct u64[2][L][N] -> out u64[2][L][N] */
int orc_ckks_rotate(size_t logn, size_t L, const orc_u64 *moduli_ext, size_t step, const orc_u64 *ct,
                    const orc_u64 *key, orc_u64 *out);
int orc_ckks_conjugate(size_t logn, size_t L, const orc_u64 *moduli_ext, const orc_u64 *ct, const orc_u64 *key,
                       orc_u64 *out);
/* mult_low_level + relinearize + rescale_inplace: out u64[2][L-1][N] */
int orc_ckks_mult_relin_rescale(size_t logn, size_t L, const orc_u64 *moduli_ext,
                                const orc_u64 *ct1, const orc_u64 *ct2,
                                const orc_u64 *key, orc_u64 *out);
/* mult_low_level + relinearize + mod_switch_inplace(t): out u64[2][L-1][N] */
int orc_bgv_mult_relin_modswitch(size_t logn, size_t L, const orc_u64 *moduli_ext,
                                 orc_u64 plain_modulus, const orc_u64 *ct1,
                                 const orc_u64 *ct2, const orc_u64 *key,
                                 orc_u64 *out);

/* ---- either side of the path ------------------- */
/* rlwe.cpp:57-72 encrypt_core, with the samples of get_rlwe_sample (rlwe.cpp:33-55) supplied by the caller:
 * noise int64[N] = the rounded Gaussian integers (sampling.cpp:60-88 lifts them per modulus and transforms),
 * c1 u64[L][N] uniform NTT-form words, pt u64[L][N] coefficient form, sk u64[L][N] NTT form -> ct u64[2][L][N] */
int orc_rlwe_encrypt_core(size_t logn, size_t L, const orc_u64 *moduli, const int64_t *noise, const orc_u64 *c1,
                          const orc_u64 *pt, const orc_u64 *sk, orc_u64 *ct);
/* rlwe.cpp:74-81 decrypt_core: pt u64[L][N] = strict(INTT(c0 + c1*sk)) */
int orc_rlwe_decrypt_core(size_t logn, size_t L, const orc_u64 *moduli, const orc_u64 *ct, const orc_u64 *sk,
                          orc_u64 *pt);
/* rns_transform.cpp:11-37 after the reduce_strict of :113: in u64[N] mod old_modulus -> out u64[L][N] */
void orc_rns_base_from_single(size_t n, orc_u64 old_modulus, size_t L, const orc_u64 *new_moduli, const orc_u64 *in,
                              orc_u64 *out);
/* rns_transform.cpp:39-84, small-coefficient branch (after the reduce_strict of :113): in u64[L][N] -> out u64[N].
 * returns 1 when every coefficient is small (out valid), 0 when the CRT branch (:86-104, BigInt) would run */
int orc_rns_base_to_single_small(size_t n, size_t L, const orc_u64 *old_moduli, orc_u64 new_modulus,
                                 const orc_u64 *in, orc_u64 *out);

/* rns_transform.cpp:106-127 with a single new modulus, BOTH branches (:39-84 small coefficients, :86-104 CRT
 * composition with the reference's big integers): in u64[L][N] (lazy allowed) -> out u64[N].  L <= 16. */
void orc_rns_base_to_single(size_t n, size_t L, const orc_u64 *old_moduli, orc_u64 new_modulus, const orc_u64 *in,
                            orc_u64 *out);

/* ---- digests / generators shared by tests, fixtures and bench ------- */
orc_u64 orc_fnv1a64(const void *bytes, size_t nbytes);
/* x[i] = splitmix64 stream (state advanced per word) mod q (q==0: raw) */
void orc_splitmix_fill(orc_u64 *state, orc_u64 q, size_t n, orc_u64 *x);

#ifdef __cplusplus
}
#endif
#endif
