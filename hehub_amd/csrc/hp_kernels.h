// hp_kernels.h -- host-callable launchers of the HIP kernels (one per kernel family).
// Every launcher enqueues on `stream` and returns the hipError_t of the launch.
#pragma once
#include "hp_device.h"

#define HP_MAX_LIMBS 32

// ---- transform jobs -----------------------------------------------------------
// A transform launch processes W independent limb transforms ("work items").
// Items are numbered modulus-major so that neighbouring items share a twiddle
// table; hp_xcd_remap() then gives each XCD a contiguous slice of them.
enum HpNttMode : int {
    HP_NTT_BATCH = 0,   // rows [P][L][N]: item w = k*P + p         -> src/dst row p*L + k, limb k
    HP_NTT_HKS = 2,     // hybrid key switch (extension): in-place transforms of the lifted digits [P][nd][E][N]; one item per
                        //   (modulus m, digit d, polynomial p) with m outside digit d, modulus-major, hole-free
    HP_NTT_SPREAD = 1,  // digit spread (rgsw.cpp:108-119): one item per (k, j != k, p), k in [0,L], numbered modulus-major
                        //   then digit then polynomial without holes, or by groups of moduli (pair_moduli, hp_ntt_job.h): src = coef row p*L + j,
                        //   dst = digit row (p*L + j)*(L+1) + k, limb k; a launch may cover only the moduli
                        //   k_first .. k_first + kc - 1 (W = their item count)
};

struct HpNttJob {
    const HpLimb *limbs;  // plan (device)
    const u64 *src;
    u64 *dst;
    u32 logn;
    u32 L;          // limbs per polynomial
    u32 P;          // polynomials
    u32 src_pstride;  // HP_NTT_BATCH / HP_NTT_LAST: rows (limbs) between consecutive polynomials of src
    u32 dst_pstride;  // HP_NTT_BATCH: same for dst
    u32 src_kstride;  // HP_NTT_BATCH: rows between consecutive limbs of src (1; 0 = every limb reads the same row)
    u32 W;          // work items
    u32 k_first;    // HP_NTT_SPREAD: first output modulus of the launch
    u32 pair_moduli;  // HP_NTT_SPREAD, whole launches only: G > 0 numbers the items by groups of G consecutive moduli so that
                      // the G transforms of one source row run in neighbouring workgroups (the row is re-read from L2)
    u32 hks_nd, hks_E, hks_alpha;   // HP_NTT_HKS: digits, moduli of the extended chain, limbs per digit (L = ciphertext moduli)
    u32 pack_mask;    // HP_NTT_SPREAD, tiled kernels: bit k set = the digit rows of output modulus k are written in the 48-bit
                      // packed row format (hp_device.h: HP_PACK48) that the inner-product kernel reads back; set by the engine only
                      // when every word of those rows is provably below 2^48 (hp_api_scheme.cpp: spread_pack_mask)
    u32 pack40_mask;  // level A only (hp_ntt_a.hip): bit k set = those rows in the 40-bit packed format HP_PACK40 instead (hp_device.h)
    int mode;
    int inverse;
    int strict;     // inverse only: reduce_strict epilogue (ntt.h:88-92)
    // inverse only: multiply by a per-launch scalar (s, s') with the Harvey
    // multiplication BEFORE the strict reduction (mod_switch.cpp:49-50); s_h == 0 and s == 0 -> off
    u64 post_scalar, post_scalar_h;
    int use_post_scalar;
    // parity level A (opt-in, hp_ctx_set_parity_level): non-NULL = the launch goes to the FP64 residue kernels of hp_ntt_a.hip,
    // which return CANONICAL residues (equal to reduce_strict of the level-B words); post_scalar / post_scalar_h then hold the
    // bit patterns of the doubles (s, RN(s / q))
    const HpLimbA *limbs_a;
    u32 src_words;  // level A, HP_NTT_SPREAD: the source rows are WORDS (strict coefficient rows a caller supplied: the limb-range stage
                    // hp_dev_ks_inner_range_strict), not the doubles k_ntt_inv_a writes with dst_f64 for the launch of its own call
    u32 dst_f64;    // level A, inverse: the output rows are the doubles themselves, not words -- only for rows that feed a level-A
                    // HP_NTT_SPREAD launch (k_ntt_fwd_a<., true> reads doubles), never for rows a caller sees
};

hipError_t hp_launch_ntt_generic(const HpNttJob &job, hipStream_t stream);
// fast path: logn in [11,15]; returns hipErrorNotSupported otherwise
hipError_t hp_launch_ntt_fast(const HpNttJob &job, hipStream_t stream);
// a limb split over N / 2048 workgroups and two launches (hp_ntt_split.hip): the latency path for launches with few limbs; logn in
// [12, 16], plain u64 rows, level B; returns hipErrorNotSupported otherwise
hipError_t hp_launch_ntt_split(const HpNttJob &job, hipStream_t stream);
// the same tiling with FP64 residue butterflies (job.limbs_a; HP_NTT_BATCH and HP_NTT_SPREAD; the inverse is always strict)
hipError_t hp_launch_ntt_a(const HpNttJob &job, hipStream_t stream);

// ---- coefficient-wise kernels on [rows][n], limb of a row = row % L ---------------
enum HpBinOp : int { HP_ADD = 0, HP_SUB = 1, HP_MUL = 2 };
hipError_t hp_launch_poly_binary(int op, const HpLimb *limbs, u32 L, u32 n, u32 rows, const u64 *a,
                                 const u64 *b, u64 *out, hipStream_t stream);
struct HpScalars {
    u64 s[HP_MAX_LIMBS];
    u64 sh[HP_MAX_LIMBS];
};
hipError_t hp_launch_poly_scalar_mul(const HpLimb *limbs, const HpScalars &sc, u32 L, u32 n, u32 rows,
                                     const u64 *a, u64 *out, hipStream_t stream);
hipError_t hp_launch_poly_strict(const HpLimb *limbs, u32 L, u32 n, u32 rows, u64 *x, hipStream_t stream);
// out = in, `words` u64 (16-byte aligned rows; an odd last word goes through hipMemcpyAsync)
hipError_t hp_launch_copy(size_t words, const u64 *in, u64 *out, hipStream_t stream);
// rows of a polynomial in separate blocks of registered host memory (device-visible addresses), moved by one kernel
#define HP_HOST_ROWS_MAX 64
struct HpHostRows {
    u64 *p[HP_HOST_ROWS_MAX];
};
hipError_t hp_launch_host_rows(bool to_host, const HpHostRows &rows, u32 count, size_t words, u64 *dev, hipStream_t stream);
hipError_t hp_launch_gather(const u32 *perm, u32 n, u32 rows, const u64 *in, u64 *out, hipStream_t stream);
hipError_t hp_launch_reverse(u32 n, u32 rows, const u64 *in, u64 *out, hipStream_t stream);
// several ciphertexts moved in one launch: ciphertext b = polynomials src[b][0], src[b][1] (u64[L][N] each, anywhere), map perm[b]
// (NULL: involution) -> out u64[count][2][L][N]
#define HP_GATHER_TABLE_MAX 32
struct HpGatherTable {
    const u64 *src[HP_GATHER_TABLE_MAX][2];
    const u32 *perm[HP_GATHER_TABLE_MAX];
};
hipError_t hp_launch_gather_many(const HpGatherTable &tab, u32 count, u32 n, u32 L, u64 *out, hipStream_t stream);

// single-vector kernels behind the drop-in mod_arith entry points
enum HpVecOp : int {
    HP_V_BARRETT_LAZY = 0,
    HP_V_BARRETT = 1,
    HP_V_STRICT = 2,
    HP_V_MUL_HYBRID = 3,
    HP_V_MUL_BARRETT = 4,
    HP_V_MONTGOMERY128 = 5
};
struct HpVecConsts {
    u64 q, mqinv, r64, r64h, barrett_c, c128_hi, c128_lo;
};
hipError_t hp_launch_vec(int op, const HpVecConsts &c, size_t n, const u64 *a, const u64 *b, u64 *out,
                         hipStream_t stream);

// ---- scheme-level kernels ---------------------------------------------------------
// ckks/arith.cpp:55-62: ct1, ct2 [P][2][L][n] -> quad [P][3][L][n]
// only limbs [k_first, k_first + kc) are computed (kc = L: all)
hipError_t hp_launch_tensor(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 n, u32 P, const u64 *ct1,
                            const u64 *ct2, u64 *quad, hipStream_t stream);
// the same with the operands by address: polynomials ct1[0], ct1[1], ct2[0], ct2[1] (u64[L][N] each) of pair p at rows.p[p][0..3]
// a chain of lazy sums / differences per polynomial: rows.p[poly * terms + j] = term j of polynomial poly (u64[L][n]), bit j of neg: subtracted
#define HP_FOLD_ROWS_MAX 64
#define HP_FOLD_TERMS_MAX 32
struct HpFoldRows {
    const u64 *p[HP_FOLD_ROWS_MAX];
    u32 neg;
};
hipError_t hp_launch_poly_fold(const HpLimb *limbs, u32 L, u32 n, u32 polys, u32 terms, const HpFoldRows &rows, u64 *out, hipStream_t stream);
#define HP_TENSOR_ROWS_MAX 64
struct HpTensorRows {
    const u64 *p[HP_TENSOR_ROWS_MAX][4];
};
hipError_t hp_launch_tensor_rows(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 n, u32 P, const HpTensorRows &rows, u64 *quad,
                                 hipStream_t stream);
// rgsw.cpp:121-153: digits [P][L][L+1][n] (diagonal taken from pt [P][L][n]), key [L][2][L+1][n]
//   -> out [P][2][L+1][n]
// only output moduli [k_first, k_first + kc) of the L+1 are computed (kc = L+1: all)
// key_Le: limbs per key polynomial (L+1, or more for a key generated at a higher level: its last column is the special prime)
// pack_mask: bit k set = the digit rows of output modulus k are in the 48-bit packed row format (needs P >= 2)
// pack40_mask: bit k set = in the 40-bit offset format of parity level A (HP_PACK40; takes precedence)
hipError_t hp_launch_ks_inner(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 key_Le, u32 n, u32 P, const u64 *digits,
                              const u64 *pt, u32 pt_pstride, const u64 *key, u64 *out, u32 pack_mask, u32 pack40_mask,
                              hipStream_t stream);
// the same inner product when every ciphertext has its own key (hp_dev_ckks_rotate_many): key addresses as kernel arguments
#define HP_KEY_TABLE_MAX 32
struct HpKeyTable {
    const u64 *p[HP_KEY_TABLE_MAX];
};
hipError_t hp_launch_ks_inner_many(const HpLimb *limbs, u32 L, u32 k_first, u32 kc, u32 key_Le, u32 n, u32 P, const u64 *digits,
                                   const u64 *pt, u32 pt_pstride, const HpKeyTable &keys, u64 *out, hipStream_t stream);

// drop-last-prime helpers (rescaling.cpp:46-75 / mod_switch.cpp:45-77)
struct HpDropConsts {
    u64 q_last, half_q_last;
    u64 r[HP_MAX_LIMBS];       // q_last mod q_k
    u64 inv[HP_MAX_LIMBS];     // q_last^{-1} mod q_k (already reduced mod q_k)
    u64 inv_h[HP_MAX_LIMBS];
    int bgv;
    u64 t[HP_MAX_LIMBS], t_h[HP_MAX_LIMBS];          // plain_modulus mod q_k      (mod_switch.cpp:70)
    u64 qlt[HP_MAX_LIMBS], qlt_h[HP_MAX_LIMBS];      // (q_last mod t) mod q_k     (mod_switch.cpp:76)
};
// Fused form for the tiled transform (rescaling.cpp:54-74 / mod_switch.cpp:52-76 in ONE launch): the forward
// NTT of the remainder limbs reads the strict last-limb coefficients, applies Barrett + centring (+ *t) while
// loading, and finishes with out = ((x - NTT(rem)) * inv) [* (q_last mod t)] [+ addend] while storing.
struct HpDropArgs {
    HpDropConsts dc;
    int small_rem;         // 1: q_last <= 2 q_k for every limb of the launch: the remainder of c < q_last is c - [c >= q_k] q_k (the canonical
                           //    residue either way; saves the Barrett quotient).  Honoured by the compile-time flavours only.
    int raw_input;         // 1: the transform's input rows already are the per-limb remainders (hybrid key switch): no
                           //    Barrett / centring prologue
    u32 out_stride;        // limbs between consecutive polynomials of out (L - 1 for a plain drop)
    int fin_on;            // 1: one more per-limb multiplication AFTER the addend (hybrid key switch: merged rescale)
    u64 fin[HP_MAX_LIMBS], fin_h[HP_MAX_LIMBS];
    const u64 *comb;       // non-NULL (with raw_input): input = src + comb_mul[k] * centre_k(comb[p2]), comb [P2][N] strict modulo 2*comb_half+1
    u64 comb_half, comb_r[HP_MAX_LIMBS], comb_mul[HP_MAX_LIMBS], comb_mul_h[HP_MAX_LIMBS];
    u64 q2_last, half_q2_last;   // level A, two drops in one launch (hp_ntt_a.hip: DropPre2A): the second modulus dropped; comb = its
                                 // strict coefficient rows [P2][N]; per limb (t, t_h) = m_k, (comb_mul, comb_mul_h) = m2_k, (inv, inv_h) = A_k,
                                 // (qlt, qlt_h) = B_k, all as bit patterns of doubles (v, RN(v / q_k))
    const u64 *x;          // [P2][L][n]: polynomial p2 at x + p2*L*n, limb k at + k*n
    u32 L;                 // limbs of x (the last one is being dropped)
    const u64 *addend;     // optional [.][.][n]: row (p2>>1)*add_ct_stride + (p2&1)*add_poly_stride + k
    u32 add_poly_stride, add_ct_stride;
    u32 add_mask;          // bit h set: polynomial h of each ciphertext gets the addend (relinearize 3, rotate 1)
    u64 *out;              // [P2][L-1][n]
};
// job: HP_NTT_BATCH over L-1 limbs and P2 polynomials with src = clast [P2][n] (src_pstride 1, src_kstride 0)
hipError_t hp_launch_ntt_fast_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream);
// the same for a launch of few limbs, split over N / 2048 workgroups per limb (hp_ntt_split.hip); job.dst = scratch rows [P2][kc][n]
hipError_t hp_launch_ntt_split_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream);
// level A: dc.q_last / half_q_last and the pairs (inv, inv_h), (t, t_h), (qlt, qlt_h) hold bit patterns of doubles (v, RN(v / q_k));
// r, small_rem, raw_input, fin, comb are not used; output rows are canonical residues
hipError_t hp_launch_ntt_a_drop(const HpNttJob &job, const HpDropArgs &da, hipStream_t stream);
// level A, inverse of ONE limb (job.L == 1) whose input is A * src + add and whose output has K * centre(cprev) subtracted
// (hp_ntt_a.hip: ntt_inv_a_body MIX); every constant as the bit pattern of a double, (v, RN(v / q)) for the multipliers
struct HpInvMixArgs {
    const u64 *add;            // addend rows already at the limb: polynomial p at (p >> 1) * add_ct_stride + (p & 1) * add_poly_stride limbs
    u32 add_poly_stride, add_ct_stride;
    u64 A, A_h;
    const u64 *cprev;          // [P][N] strict coefficients modulo prev_q
    u64 prev_q, prev_half;
    u64 K, K_h;
};
hipError_t hp_launch_ntt_a_inv_mix(const HpNttJob &job, const HpInvMixArgs &mx, hipStream_t stream);

// clast [P2][n] (strict coefficients of the last limb) -> rem [P2][L-1][n]
hipError_t hp_launch_drop_rem(const HpLimb *limbs, const HpDropConsts &dc, u32 Lm1, u32 n, u32 P2,
                              const u64 *clast, u64 *rem, hipStream_t stream);
// x [P2][L][n] (first L-1 limbs used), rem [P2][L-1][n] (NTT form), optional addend [P2][addL][n]
//   -> out [P2][L-1][n]:  out = ((x - rem) * inv) [* qlt]  [+ addend]
// kc limbs per polynomial (L-1: all); for a limb range the caller shifts limbs / dc / x / addend / out to its first limb
hipError_t hp_launch_drop_fin(const HpLimb *limbs, const HpDropConsts &dc, u32 L, u32 kc, u32 n, u32 P2, const u64 *x,
                              const u64 *rem, const u64 *addend, u32 add_poly_stride, u32 add_ct_stride,
                              u32 add_mask, u64 *out, hipStream_t stream);

// constants of the CRT branch of rns_base_transform many -> one (device memory; built by the engine)
#define HP_CRT_MAX_LIMBS 16
struct HpCrtConsts {
    u64 t, q_mod_t;                                                  // new modulus, (q_0 ... q_{L-1}) mod t
    u64 inv[HP_CRT_MAX_LIMBS][HP_CRT_MAX_LIMBS], inv_h[HP_CRT_MAX_LIMBS][HP_CRT_MAX_LIMBS];   // [b][a], b < a: q_b^-1 mod q_a (+ Harvey word)
    u64 half[HP_CRT_MAX_LIMBS];                                      // mixed-radix digits of floor(Q/2)
    u64 pref[HP_CRT_MAX_LIMBS], pref_h[HP_CRT_MAX_LIMBS];            // q_0 ... q_{a-1} mod t (+ Harvey word)
};
// out: polynomial p at out + p*out_pstride*n; not_small == NULL: every polynomial takes the CRT composition
hipError_t hp_launch_base_to_single_crt(const HpLimb *limbs, const HpCrtConsts *cc, u32 L, u32 n, u32 P, const u64 *in, u64 *out,
                                        u32 out_pstride, const u32 *not_small, hipStream_t stream);

// ---- either side of the path ---------------------------------------
// noise [P][n] (int64) -> out rows p*out_pstride + k: the per-modulus lift of sampling.cpp:77-83
hipError_t hp_launch_lift_noise(const HpLimb *limbs, u32 L, u32 n, u32 P, const long long *noise, u64 *out, u32 out_pstride,
                                hipStream_t stream);
// ct[p][0] holds NTT(lift(noise)) on entry; c1, ptn [P][L][n], sk [L][n] -> ct [P][2][L][n]   (rlwe.cpp:52,70)
hipError_t hp_launch_enc_fin(const HpLimb *limbs, u32 L, u32 n, u32 P, const u64 *c1, const u64 *sk, const u64 *ptn, u64 *ct,
                             hipStream_t stream);
// out [P][L][n] = c0 + c1*sk   (rlwe.cpp:76)
hipError_t hp_launch_dec_fma(const HpLimb *limbs, u32 L, u32 n, u32 P, const u64 *ct, const u64 *sk, u64 *out,
                             hipStream_t stream);
// rns_transform.cpp:11-37 / :39-84; limbs = plan of the NEW (from_single) / OLD (to_single) moduli
hipError_t hp_launch_base_from_single(const HpLimb *limbs, u64 old_q, u32 L, u32 n, u32 P, const u64 *in, u64 *out,
                                      hipStream_t stream);
hipError_t hp_launch_base_to_single(const HpLimb *limbs, u32 L, u32 n, u32 P, u64 new_q, const u64 *in, u64 *out,
                                    u32 *not_small, hipStream_t stream);

// ---- hybrid key switch (extension; hp_hks.hip) ---------------------------------------------------
#define HP_HKS_MAX_ALPHA 8
#define HP_HKS_MAX_DIGITS 16
struct HpHksConsts {   // device memory, built by the engine per (moduli, k, alpha)
    u32 L, k, alpha, nd, E;   // ciphertext moduli, special primes, limbs per digit, digits, L + k
    // Garner inverses inside digit d: [d][b][a], b < a: (b-th modulus of the digit)^-1 mod (a-th modulus of the digit)
    u64 inv[HP_HKS_MAX_DIGITS][HP_HKS_MAX_ALPHA][HP_HKS_MAX_ALPHA], inv_h[HP_HKS_MAX_DIGITS][HP_HKS_MAX_ALPHA][HP_HKS_MAX_ALPHA];
    // [d][m][a]: product of the first a moduli of digit d, modulo modulus m of the extended chain
    u64 pref[HP_HKS_MAX_DIGITS][HP_MAX_LIMBS][HP_HKS_MAX_ALPHA], pref_h[HP_HKS_MAX_DIGITS][HP_MAX_LIMBS][HP_HKS_MAX_ALPHA];
    u64 pinv[HP_MAX_LIMBS], pinv_h[HP_MAX_LIMBS];   // (p_0...p_{k-1})^-1 mod q_i
    // ModDown (k <= HP_HKS_MAX_ALPHA special primes): Garner inverses among them, digits of floor(P/2), and per ciphertext
    // modulus q_i the prefix products p_0...p_{a-1} mod q_i and P mod q_i
    u64 pg_inv[HP_HKS_MAX_ALPHA][HP_HKS_MAX_ALPHA], pg_inv_h[HP_HKS_MAX_ALPHA][HP_HKS_MAX_ALPHA];
    u64 p_half[HP_HKS_MAX_ALPHA];
    u64 p_pref[HP_MAX_LIMBS][HP_HKS_MAX_ALPHA], p_pref_h[HP_MAX_LIMBS][HP_HKS_MAX_ALPHA];
    u64 p_mod_q[HP_MAX_LIMBS], p_mod_q_h[HP_MAX_LIMBS];
};
// yp [P2][k][n] (strict coefficients of the special-prime part) -> rem [P2][L][n]: the exact centred value in every q_i
hipError_t hp_launch_hks_moddown(const HpLimb *limbs, const HpHksConsts *hc, u32 k, u32 n, u32 P2, const u64 *yp, u64 *rem,
                                 hipStream_t stream);
hipError_t hp_launch_hks_modup(const HpLimb *limbs, const HpHksConsts *hc, u32 alpha, u32 nd, u32 n, u32 P, const u64 *coef,
                               u64 *lifted, hipStream_t stream);
hipError_t hp_launch_hks_inner(const HpLimb *limbs, u32 L, u32 E, u32 nd, u32 alpha, u32 n, u32 P, const u64 *lifted, const u64 *pt,
                               u32 pt_pstride, const u64 *key, u64 *out, hipStream_t stream);
// merged ModDown + rescale (hp_engine.cpp: hks_mult): rem[p2][i] += (P mod q_i) * centre(c_last[p2]) for i < L-1, in the
// coefficient domain; c_last = strict coefficients modulo q_{L-1} of the relinearised limb L-1
hipError_t hp_launch_hks_combine(const HpLimb *limbs, const HpHksConsts *hc, u32 L, u32 n, u32 P2, const u64 *clast, u64 *rem,
                                 hipStream_t stream);
hipError_t hp_launch_hks_down_fin(const HpLimb *limbs, const HpHksConsts *hc, u32 L, u32 n, u32 P2, const u64 *x, const u64 *rem,
                                  const u64 *addend, u32 add_poly_stride, u32 add_ct_stride, u32 add_mask, u64 *out,
                                  hipStream_t stream);
