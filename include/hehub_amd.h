/*
 * hehub_amd.h -- C ABI of the MI355X-native RNS ring-arithmetic engine.
 *
 * This is the drop-in boundary for primihub/hehub's data-parallel hot path
 * (negacyclic NTT/INTT, coefficient-wise modular arithmetic, key-switch inner
 * product, rescale / mod-switch).  hehub has no FFI layer of its own: its seam
 * is the set of free functions in namespace hehub declared in
 *   src/fhe/common/ntt.h, mod_arith.h, rns.h, src/fhe/primitives/rgsw.h,
 *   src/fhe/ckks/ckks.h, src/fhe/bgv/bgv.h .
 * Each entry point below names the reference declaration (file:line, relative
 * to the hehub source tree) it stands in for.  The C++ shim a hehub maintainer
 * links instead of hehub's own .cpp files is hehub_amd/host/ (see
 * INTEGRATION.md).
 *
 * Conventions
 *   - plain C: pointers, sizes, integers.  No torch / STL types.
 *   - every function returns an hp_status; nothing throws across the ABI; a NULL
 *     ctx is HP_EINVAL.  hp_last_error(ctx) gives the message of the calling
 *     thread's last failure on that ctx (valid until its next hp_* call).
 *   - one hp_ctx per GPU; calls on one ctx are serialised by an internal
 *     mutex; different ctxs are independent.
 *   - "host" entry points take caller-owned host pointers, are synchronous and
 *     retain nothing.  "dev" entry points take device pointers (hipMalloc'd by
 *     the caller, by hp_dev_alloc, or by any other allocator in the process,
 *     e.g. a torch tensor's data_ptr()), enqueue on the ctx stream and return
 *     without synchronising.
 *   - device pointers must be 16-byte aligned (the kernels move 16 bytes per lane; every allocator's blocks are, and
 *     so is every limb inside them); a NULL or misaligned operand is rejected with HP_EINVAL before anything runs.
 *   - all words are uint64_t, little-endian, in the reference's lazy
 *     (redundant) representation; outputs are bit-identical to the
 *     reference's for identical inputs.
 *
 * Device layouts (row-major):
 *   polynomial batch   u64[batch][L][N]
 *   ciphertext batch   u64[batch][2][L][N]     (quadratic: [batch][3][L][N])
 *   key-switch key     u64[L][2][L+1][N]       rgsw[j][half][k], Montgomery form
 *                                               (keys.cpp:8-36, rgsw.cpp:33-55)
 */
#ifndef HEHUB_AMD_H
#define HEHUB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hp_ctx hp_ctx;

typedef enum {
    HP_OK = 0,
    HP_EINVAL = 1,       /* reference would throw std::invalid_argument */
    HP_EUNSUPPORTED = 2, /* reference throws "under development" / not built */
    HP_EHIP = 3,         /* HIP runtime failure */
    HP_ENOMEM = 4,
    HP_ELOGIC = 5,
    HP_ERANGE = 6        /* parity level A only: a kernel was handed a word that is not a lazy word of its limb (see hp_parity_level) */
} hp_status;

/* ---- engine ------------------------------------------------------------ */
int hp_ctx_create(int device, hp_ctx **out);
void hp_ctx_destroy(hp_ctx *ctx);
/* Parity level of the scheme-level pipelines ("Parity levels").
 *   HP_PARITY_B (default): every output word is the raw lazy u64 word hehub's CPU code produces, bit for bit.
 *   HP_PARITY_A (opt-in; HP_PARITY_LEVEL=A in the environment at hp_ctx_create): the key switch, drop-last-prime, relinearisation,
 *     rotation and fused mult entry points (hp_dev_ckks_rescale*, hp_dev_bgv_mod_switch, hp_dev_*_relinearize*,
 *     hp_dev_ckks_rotate* / conjugate*, hp_dev_*_mult_relin_*) return the CANONICAL residue in [0, q) of every word: congruent
 *     to hehub's word modulo its q and equal to reduce_strict (mod_arith.h:58-72) of it -- the contract of
 *     intt_negacyclic_inplace (ntt.h:88-92) extended to the whole pipeline.  hp_dev_ext_prod_montgomery* returns lazy words
 *     (< 2q, rgsw.cpp:151) with hehub's residues.  Their transforms then run on error-free FP64
 *     products (hp_ntt_a.hip: 8 instead of 16 instructions per butterfly).  Needs every modulus of the chain below 2^50, a ring
 *     degree of 2^11 .. 2^15; a call whose chain does not qualify runs at level B.  Input words must be LAZY words of their limb
 *     (below 2 q: everything hehub or this engine produces is; hehub's own transforms take any u64, ntt.cpp:155-175, and level B
 *     reproduces that).  The precondition is CHECKED: every word a level-A kernel loads from a caller's row passes a range guard
 *     (one instruction), a word above 2 q sets a sticky flag, and the next synchronising call on the family (hp_sync,
 *     hp_memcpy_d2h) returns HP_ERANGE -- the results since the previous synchronisation are then void.  At this level the fused mult entry points (hp_dev_*_mult_relin_*) also merge relinearize's
 *     mod-down with the rescale / mod switch into one transform per output limb (same residues; HP_NO_DOUBLE_DROP=1 at
 *     hp_ctx_create keeps the two launches).  The NTT / mod-arith primitives (hp_ntt_*, hp_dev_ntt_*, hp_dev_poly_*, hp_batched_*) are
 *     never affected: they stay bit-exact with ntt.cpp:145-223 / mod_arith.cpp. */
/* A second LANE on the parent's GPU (hehub's callers are loops of independent single-ciphertext operations,
 * src/circuits/linear_algebra.h:109-133, bench/benchmarks.cpp:24-35: one ciphertext fills 10 .. 100 of the 256 CUs, so independent
 * calls should overlap): a context with its own stream and scratch workspace whose calls run concurrently on the device with those
 * of its parent and siblings.  The family shares ONE lock (host-side enqueueing is serialised exactly as on a single context) and ONE
 * copy of the twiddle tables / per-chain constants / gather maps.  Knobs and parity level start as the parent's are at the fork.
 * Destroy every member with hp_ctx_destroy (any order).  hp_ctx_wait_for is the ordering primitive between lanes: everything
 * `ctx` enqueues from now on runs after everything `other` has enqueued so far -- a device-side event, no host synchronisation.
 * Buffers are the caller's: a lane that reads what another lane wrote (or reuses memory another lane still reads) needs that
 * ordering; hehub_amd/host keeps the book per device block. */
int hp_ctx_fork(hp_ctx *parent, hp_ctx **out);
int hp_ctx_wait_for(hp_ctx *ctx, hp_ctx *other);
typedef enum { HP_PARITY_B = 0, HP_PARITY_A = 1 } hp_parity_level;
int hp_ctx_set_parity_level(hp_ctx *ctx, int level);
int hp_ctx_get_parity_level(hp_ctx *ctx);
const char *hp_last_error(hp_ctx *ctx);
const char *hp_version(void);
/* enqueue on an existing hipStream_t (e.g. torch's current stream); NULL = the HIP default stream.
 * A fresh ctx uses a private non-blocking stream; hp_ctx_reset_stream goes back to it.  Switching streams keeps device
 * order: the new stream first waits (event, no host synchronisation) for everything the ctx enqueued on the previous one,
 * because the scratch workspace and the cached tables are shared by all calls of a ctx.
 * HIP graphs: the first hp_dev_* call with a new (ring degree, moduli, shape) builds tables / constants and may grow the
 * workspace (allocations and host-to-device copies); every later call with the same parameters only enqueues kernels on the
 * ctx stream, so it can be recorded with hipStreamBeginCapture on that stream and replayed (tests/test_gpu_parity.py).
 * A later call that needs a LARGER workspace drains the device, frees the old block and allocates a new one: graphs captured
 * before that hold stale scratch pointers and must be re-captured -- hp_ctx_workspace_generation() changes whenever that
 * happened (also on hp_ctx_release_workspace).  Warm up with the largest shape first to avoid it. */
int hp_ctx_set_stream(hp_ctx *ctx, void *hip_stream);
int hp_ctx_reset_stream(hp_ctx *ctx);
void *hp_ctx_get_stream(hp_ctx *ctx);
int hp_sync(hp_ctx *ctx);
/* the engine's scratch workspace grows to the largest call so far and is reused; these report / release it */
size_t hp_ctx_workspace_bytes(hp_ctx *ctx);
unsigned long hp_ctx_workspace_generation(hp_ctx *ctx);
int hp_ctx_release_workspace(hp_ctx *ctx);
int hp_dev_alloc(hp_ctx *ctx, size_t bytes, void **dptr);
int hp_dev_free(hp_ctx *ctx, void *dptr);
/* page-locked host memory for batches that cross PCIe (hp_memcpy_*, the node layer's host entry points): copies from / to it are
 * DMA at the link rate whatever the state of the pages (pageable memory measured 22 .. 48 GB/s from box to box, page-locked
 * 49 .. 52 GB/s); portable across the devices of a node */
int hp_host_alloc(hp_ctx *ctx, size_t bytes, void **hptr);
int hp_host_free(hp_ctx *ctx, void *hptr);
int hp_memcpy_h2d(hp_ctx *ctx, void *dst, const void *src, size_t bytes);
int hp_memcpy_d2h(hp_ctx *ctx, void *dst, const void *src, size_t bytes);
/* Host memory the caller owns, made DMA-able in place (hipHostRegister) -- the limbs of hehub's SmartArray pool
 * (allocator.h:19-22: blocks are recycled, never returned to the OS) can be registered once and then cross PCIe at the link rate
 * without the runtime's staging of pageable memory.  The *_async copies only enqueue on the ctx stream: the host buffer must stay
 * untouched until hp_sync() (or any synchronous hp_* call on the ctx) returns. */
int hp_host_register(hp_ctx *ctx, void *hptr, size_t bytes);
int hp_host_unregister(hp_ctx *ctx, void *hptr);
int hp_memcpy_h2d_async(hp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int hp_memcpy_d2h_async(hp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* Device memory of one context -> device memory of another (the two may sit on different GPUs of the node, on one GPU, or be the same
 * context): enqueued on DST's stream.  Order it behind the producer with hp_ctx_wait_for(dst_ctx, src_ctx) first.  Between two GPUs the
 * copy crosses one xGMI link (peer access is enabled for the pair on first use; without it the runtime stages the copy).  hehub has no
 * devices; the host layer uses this when an operand lives on another GPU than the call (hehub_amd/host: "devices"). */
int hp_memcpy_peer_async(hp_ctx *dst_ctx, void *d_dst, hp_ctx *src_ctx, const void *d_src, size_t bytes);
int hp_ctx_device(hp_ctx *ctx);   /* the HIP device ordinal the context was created on (HP_EINVAL < 0 for NULL) */
/* A polynomial whose limbs are SEPARATE registered host blocks (rns.h:15-156: one SmartArray per limb) <-> its contiguous device rows
 * u64[rows][words], by ONE kernel that reads / writes the host blocks over PCIe: 47-49 GB/s either way on an MI355X against
 * 11-17 GB/s for one DMA command per 256 KiB block (tools/ubench/ubench_pcie.hip, profiles/archive/r04_ubench_pcie.txt).  Every h_rows[r] must be
 * 16-byte aligned memory from hp_host_alloc or registered with hp_host_register; words even.  Enqueue only (see above). */
int hp_dev_store_host_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *d_src, uint64_t *const *h_rows);
int hp_dev_load_host_rows(hp_ctx *ctx, size_t rows, size_t words, uint64_t *d_dst, const uint64_t *const *h_rows);
/* The same between device rows that lie ANYWHERE (the ciphertexts of an application are separate objects, rlwe.h:27; the batch
 * entry points below want u64[batch][2][L][N]) and one packed block u64[rows][words]: d_rows is a HOST array of `rows` device
 * pointers, 16-byte aligned; words even.  One streaming kernel per 64 rows on the ctx stream. */
int hp_dev_gather_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *const *d_rows, uint64_t *d_dst);
int hp_dev_scatter_rows(hp_ctx *ctx, size_t rows, size_t words, const uint64_t *d_src, uint64_t *const *d_rows);
/* force the simple one-stage-at-a-time transform kernels (debug / cross-check) */
int hp_ctx_set_force_generic(hp_ctx *ctx, int on);
/* timing of the last profiled launch group: see hp_prof_* below */

/* ---- drop-in, host pointers (one call = one reference call) -------------- */
/* ntt.h:33   void ntt_negacyclic_inplace_lazy(size_t log_dimension, u64 modulus, u64 coeffs[]) */
int hp_ntt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t log_dimension, uint64_t modulus, uint64_t *coeffs);
/* ntt.h:64   void intt_negacyclic_inplace_lazy(size_t log_dimension, u64 modulus, u64 values[]) */
int hp_intt_negacyclic_inplace_lazy(hp_ctx *ctx, size_t log_dimension, uint64_t modulus, uint64_t *values);
/* ntt.h:102  void cache_ntt_factors_strict(u64 log_dimension, const std::vector<u64>&) */
int hp_cache_ntt_factors_strict(hp_ctx *ctx, size_t log_dimension, const uint64_t *moduli, size_t count);
/* Would the engine accept this modulus chain?  Builds (and keeps) exactly what the first real call on the chain builds -- per-limb
 * constants, and for log_dimension != 0 the twiddle tables of every modulus at that ring degree -- and returns what that call would:
 * HP_EINVAL for a modulus the transforms reject (2N does not divide q - 1, more than 59 bits: ntt.cpp:26-29,43-47), for moduli that are
 * not pairwise coprime, for an even modulus when `montgomery` is set (the products of rns.cpp:120-140 need -q^-1 mod 2^64,
 * mod_arith.cpp:49-52); HP_EUNSUPPORTED for a ring degree outside 2 .. 2^16.  log_dimension == 0: no transforms on the chain.
 * The host layer calls this when it RECORDS a call instead of running it, so that the exception comes at the call, as hehub's does. */
int hp_check_chain(hp_ctx *ctx, size_t log_dimension, const uint64_t *moduli, size_t count, int montgomery);
/* mod_arith.h:16   batched_barrett_lazy(modulus, vec_len, vec) */
int hp_batched_barrett_lazy(hp_ctx *ctx, uint64_t modulus, size_t vec_len, uint64_t *vec);
/* mod_arith.h:18   batched_barrett */
int hp_batched_barrett(hp_ctx *ctx, uint64_t modulus, size_t vec_len, uint64_t *vec);
/* mod_arith.h:58   batched_reduce_strict */
int hp_batched_reduce_strict(hp_ctx *ctx, uint64_t modulus, size_t vec_len, uint64_t *vec);
/* mod_arith.h:27   batched_mul_mod_hybrid_lazy(modulus, vec_len, in1, in2, out) */
int hp_batched_mul_mod_hybrid_lazy(hp_ctx *ctx, uint64_t modulus, size_t vec_len, const uint64_t *in1,
                                   const uint64_t *in2, uint64_t *out);
/* mod_arith.h:41   batched_mul_mod_barrett_lazy */
int hp_batched_mul_mod_barrett_lazy(hp_ctx *ctx, uint64_t modulus, size_t vec_len, const uint64_t *in1,
                                    const uint64_t *in2, uint64_t *out);
/* mod_arith.h:55   batched_montgomery_128_lazy(modulus, len, const u128 in[], u64 out[]); in = {lo,hi} pairs */
int hp_batched_montgomery_128_lazy(hp_ctx *ctx, uint64_t modulus, size_t len, const uint64_t *in128,
                                   uint64_t *out);

/* ---- device-resident batches (throughput path) --------------------------- */
/* ntt.h:41-51 / :72-82 applied to every polynomial of a batch u64[batch][L][N], in place.
 * strict != 0 on the inverse also applies reduce_strict (ntt.h:88-92). */
int hp_dev_ntt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x);
int hp_dev_intt(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x,
                int strict);
/* The same transforms as RESIDUES (parity level A as explicit entry points, whatever the context's level; hp_ntt_a.hip: error-free
 * FP64 butterflies, 8 instead of 16 instructions each): in place, every output word is the canonical residue in [0, q).
 *   hp_dev_ntt_residues:  word == (ntt.cpp:145-176's lazy word) mod q, i.e. reduce_strict of it where that one is below 2q
 *   hp_dev_intt_residues: the words of intt_negacyclic_inplace (ntt.h:88-92 = lazy inverse + reduce_strict), bit for bit
 * Input words must be lazy words of their limb (below 2 q; checked: HP_ERANGE from the next hp_sync / hp_memcpy_d2h, see
 * hp_parity_level); N = 2^11 .. 2^15 and every modulus below 2^50, else HP_EUNSUPPORTED.  hp_dev_ntt / hp_dev_intt above stay bit-exact with the reference's lazy words. */
int hp_dev_ntt_residues(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x);
int hp_dev_intt_residues(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *d_x);
/* rns.cpp:58-87 / :89-118 / :120-140 / :142-171 on u64[batch][L][N].  d_self may alias d_out. */
int hp_dev_poly_add(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                    const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out);
int hp_dev_poly_sub(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                    const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out);
/* A chain of operator+= / operator-= (rns.cpp:58-87 / :89-118 applied term after term, e.g. the accumulate of the diagonal loop in
 * src/circuits/linear_algebra.h:117-121) folded into one pass: d_out[p] = ((x_0 op_1 x_1) op_2 x_2) ... op_{terms-1} x_{terms-1} with
 * x_j = d_rows[p * terms + j] (u64[L][N] each, anywhere on the device, 16-byte aligned; d_rows itself is a HOST array) and op_j = -=
 * where negate[j] != 0 (negate[0] is ignored).  Each step is the lazy sum / difference of the single call, in the calls' order: the
 * words of the chain of single calls; the intermediate sums never cross HBM.  d_out u64[polys][L][N]: row p may be polynomial p's own
 * x_0 (the chain then runs in place); an output row that overlaps any other input row is refused (HP_EINVAL). */
int hp_dev_poly_fold_rows(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t polys, size_t terms,
                          const uint8_t *negate, const uint64_t *const *d_rows, uint64_t *d_out);
int hp_dev_poly_mul(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                    const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out);
/* per-limb scalars (operator*=(vector<u64>)); pass the same value L times for operator*=(u64) */
int hp_dev_poly_scalar_mul(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                           const uint64_t *rns_scalar, const uint64_t *d_a, uint64_t *d_out);
/* mod_arith.h:65-72 */
int hp_dev_poly_reduce_strict(hp_ctx *ctx, size_t n, size_t L, const uint64_t *moduli, size_t batch,
                              uint64_t *d_x);
/* allocator.h:113-118 (SmartArray(const SmartArray&): a deep copy) for words that live on the device: d_dst[i] = d_src[i],
 * ranges must not overlap.  One streaming kernel on the ctx stream; bench.py also times it as the measured HBM stream ceiling. */
int hp_dev_copy(hp_ctx *ctx, size_t words, const uint64_t *d_src, uint64_t *d_dst);
/* permutation.cpp:59-75 / :28-57 (NTT-form gathers) */
int hp_dev_poly_involution(hp_ctx *ctx, size_t logn, size_t L, size_t batch, const uint64_t *d_in,
                           uint64_t *d_out);
int hp_dev_poly_cycle(hp_ctx *ctx, size_t logn, size_t L, size_t batch, size_t step, const uint64_t *d_in,
                      uint64_t *d_out);

/* ckks/arith.cpp:55-62, bgv/arith.cpp:59-69: ct1, ct2 u64[batch][2][L][N] -> u64[batch][3][L][N] */
int hp_dev_mult_low_level(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                          const uint64_t *d_ct1, const uint64_t *d_ct2, uint64_t *d_quad);
/* rgsw.h:51  RlweCt ext_prod_montgomery(const RlwePt&, const RgswCt&):
 * pt u64[batch][L][N] (NTT form), key u64[L][2][L+1][N] -> out u64[batch][2][L+1][N].
 * moduli_ext = q_0..q_{L-1}, p. */
int hp_dev_ext_prod_montgomery(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                               const uint64_t *d_pt, const uint64_t *d_key, uint64_t *d_out);
/* ckks.h:313 rescale_inplace(ct, 1) -> rescaling.cpp:14-78: ct u64[batch][2][L][N] -> u64[batch][2][L-1][N] */
int hp_dev_ckks_rescale(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                        const uint64_t *d_ct, uint64_t *d_out);
/* bgv.h:167 mod_switch_inplace(ct, 1) -> mod_switch.cpp:13-78 */
int hp_dev_bgv_mod_switch(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus,
                          size_t batch, const uint64_t *d_ct, uint64_t *d_out);
/* ckks.h relinearize (ckks/arith.cpp:64-73): quad u64[batch][3][L][N] -> u64[batch][2][L][N] */
int hp_dev_ckks_relinearize(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                            const uint64_t *d_quad, const uint64_t *d_key, uint64_t *d_out);
/* bgv.h:159 relinearize (bgv/arith.cpp:71-79).  inner_plain_modulus = 1 reproduces the reference. */
int hp_dev_bgv_relinearize(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext,
                           uint64_t inner_plain_modulus, size_t batch, const uint64_t *d_quad,
                           const uint64_t *d_key, uint64_t *d_out);
/* ckks.h:284 rotate(ct, rot_key, step) / ckks.h:282 conjugate(ct, conj_key)  (ckks/arith.cpp:75-93):
 * ct u64[batch][2][L][N], key u64[L][2][L+1][N] -> out u64[batch][2][L][N] */
int hp_dev_ckks_rotate(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t step,
                       const uint64_t *d_ct, const uint64_t *d_rot_key, uint64_t *d_out);
int hp_dev_ckks_conjugate(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                          const uint64_t *d_ct, const uint64_t *d_conj_key, uint64_t *d_out);
/* ckks.h:270 mult(ct1, ct2, relin_key) followed by ckks.h:313 rescale_inplace:
 * ct1, ct2 u64[batch][2][L][N] -> out u64[batch][2][L-1][N] */
int hp_dev_ckks_mult_relin_rescale(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext,
                                   size_t batch, const uint64_t *d_ct1, const uint64_t *d_ct2,
                                   const uint64_t *d_key, uint64_t *d_out);
/* bgv::mult_low_level + bgv::relinearize + bgv::mod_switch_inplace (bgv.h:150-167) */
int hp_dev_bgv_mult_relin_modswitch(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext,
                                    uint64_t plain_modulus, size_t batch, const uint64_t *d_ct1,
                                    const uint64_t *d_ct2, const uint64_t *d_key, uint64_t *d_out);
/* Extension (SURVEY.md 8c caveat 1): the same pipeline with the inner switch of bgv::relinearize run with the plain modulus t
 * instead of hehub's 1, so that the key-switched term survives and the product decrypts.  The key must be made for it: row j
 * encrypts (p mod q_j) * ((p mod t)^-1 mod q_j) * s^2 in limb j with noise lifted by t (tests/test_extensions.py has the
 * recipe); with hehub's own keys use the parity entry point above.  Equals hp_dev_bgv_relinearize(inner_plain_modulus = t)
 * between hp_dev_mult_low_level and hp_dev_bgv_mod_switch, word for word. */
int hp_dev_bgv_mult_relin_modswitch_t(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext,
                                      uint64_t plain_modulus, size_t batch, const uint64_t *d_ct1,
                                      const uint64_t *d_ct2, const uint64_t *d_key, uint64_t *d_out);

/* ---- either side of the path: what a pipeline needs to keep ciphertexts on the device ---- */
/* rlwe.cpp:57-72 encrypt_core with the samples of get_rlwe_sample supplied by the caller (sampling stays on the host):
 * noise int64[batch][N] rounded Gaussian integers (lifted per modulus and transformed as sampling.cpp:76-86 does),
 * c1 u64[batch][L][N] uniform NTT-form words, pt u64[batch][L][N] coefficient form, sk u64[L][N] NTT form
 * -> ct u64[batch][2][L][N] */
int hp_dev_rlwe_encrypt_core(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                             const int64_t *d_noise, const uint64_t *d_c1, const uint64_t *d_pt, const uint64_t *d_sk,
                             uint64_t *d_ct);
/* rlwe.h decrypt_core (rlwe.cpp:74-81): pt u64[batch][L][N] = strict(INTT(c0 + c1*sk)) */
int hp_dev_rlwe_decrypt_core(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch,
                             const uint64_t *d_ct, const uint64_t *d_sk, uint64_t *d_pt);
/* rns_transform.h rns_base_transform, one modulus -> many (rns_transform.cpp:113 + :11-37):
 * in u64[batch][N] (mod old_modulus, lazy allowed) -> out u64[batch][L][N] */
int hp_dev_rns_base_from_single(hp_ctx *ctx, size_t n, uint64_t old_modulus, size_t L, const uint64_t *new_moduli,
                                size_t batch, const uint64_t *d_in, uint64_t *d_out);
/* rns_base_transform, many -> one, small-coefficient branch (rns_transform.cpp:113 + :39-84):
 * in u64[batch][L][N] -> out u64[batch][N]; d_not_small u32[batch] is set non-zero for a polynomial whose coefficients
 * are not all small -- the reference then composes by CRT with BigInt (:86-104), which stays on the host */
int hp_dev_rns_base_to_single_small(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, uint64_t new_modulus,
                                    size_t batch, const uint64_t *d_in, uint64_t *d_out, uint32_t *d_not_small);

/* rns_base_transform, many -> one, COMPLETE (rns_transform.cpp:106-127 with one new modulus): per polynomial the
 * small-coefficient branch when all its coefficients are small, otherwise the CRT composition the reference does with
 * big integers (:86-104) -- here with mixed-radix word arithmetic, same values (including its returning new_modulus
 * itself when new_modulus divides Q - x).  At most 16 odd pairwise-coprime old moduli, new_modulus < 2^62. */
int hp_dev_rns_base_to_single(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, uint64_t new_modulus,
                              size_t batch, const uint64_t *d_in, uint64_t *d_out);

/* ---- limb-range stages: the limb-sharded ("latency") mode across GPUs -------------------------------------
 * one ciphertext operation is cut by OUTPUT MODULUS.  Rank g owns a contiguous range
 * [k0,k1) of the extended moduli q_0..q_{L-1},p.  Every buffer keeps the single-GPU layout of the entry points
 * above; a stage reads what it needs and writes only the limbs it is asked for, so ranks owning disjoint ranges
 * fill disjoint parts of identical buffers and exchange them between stages (hehub_amd/sharded.py: all-gather of
 * the coefficient-form digit limbs, broadcast of the one coefficient limb a drop needs).  Every sum over the
 * digits j is still formed on ONE GPU in the reference's order, so results stay bit-identical (Level B). */
/* ckks/arith.cpp:55-62 for the limbs k in [k0,k1) of L */
int hp_dev_mult_low_level_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t batch, size_t k0,
                                size_t k1, const uint64_t *d_ct1, const uint64_t *d_ct2, uint64_t *d_quad);
/* rgsw.cpp:103-105 for the digits j in [j0,j1) of L: coef[p][j] = strict(INTT(pt[p][j])).
 * pt: polynomial p starts pt_pstride limbs after polynomial p-1; coef u64[batch][L][N] */
int hp_dev_ks_coef_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t j0,
                         size_t j1, const uint64_t *d_pt, size_t pt_pstride, uint64_t *d_coef);
/* rgsw.cpp:108-153 for the output moduli k in [k0,k1) of L+1: needs ALL L limbs of coef, writes out[p][half][k] */
int hp_dev_ks_inner_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0,
                          size_t k1, const uint64_t *d_coef, const uint64_t *d_pt, size_t pt_pstride,
                          const uint64_t *d_key, uint64_t *d_out);
/* The same when the caller vouches that every row of coef holds strict residues (the rows were written by hp_dev_ks_coef_range, here or
 * on another rank): the digit rows of the workspace may then be packed (HP_PACK48), and at parity level A the digit transforms run on
 * the FP64 kernels (same residues; the output words are then lazy words with hehub's residues, as hp_dev_ext_prod_montgomery's). */
int hp_dev_ks_inner_range_strict(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch, size_t k0,
                                 size_t k1, const uint64_t *d_coef, const uint64_t *d_pt, size_t pt_pstride,
                                 const uint64_t *d_key, uint64_t *d_out);
/* rescaling.cpp:47-50 / mod_switch.cpp:47-50 (plain_modulus 0 = CKKS): clast[p2] = strict(INTT_{q_last}(x[p2][L-1]))
 * for P2 polynomials of L limbs; run by the owner of the limb that is being dropped.  clast u64[P2][N] */
int hp_dev_drop_coeffs(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus, size_t P2,
                       const uint64_t *d_x, uint64_t *d_clast);
/* rescaling.cpp:54-74 / mod_switch.cpp:52-76 for the limbs k in [k0,k1) of the L-1 that stay, given clast:
 * out[p2][k] = ((x[p2][k] - NTT_k(rem_k(clast[p2]))) * q_last^-1) [* (q_last mod t)] [+ addend], x u64[P2][L][N],
 * out u64[P2][L-1][N]; addend row of polynomial p2, limb k: (p2>>1)*add_ct_stride + (p2&1)*add_poly_stride + k,
 * applied to polynomial h = p2&1 of each ciphertext when bit h of add_mask is set (NULL: none) */
int hp_dev_drop_apply_range(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus,
                            size_t P2, size_t k0, size_t k1, const uint64_t *d_x, const uint64_t *d_clast,
                            const uint64_t *d_addend, size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask,
                            uint64_t *d_out);
/* The same when the caller vouches that every word of clast is below q_last (the rows were written by hp_dev_drop_coeffs):
 * where q_last <= 2 q_k the remainder of rescaling.cpp:54-58 is then one conditional subtraction instead of a Barrett quotient --
 * the canonical residue either way, so the words are identical.  hp_dev_drop_apply_range makes no such assumption.
 * The _strict stages (this one, hp_dev_ks_inner_range_strict) and hp_dev_ks_coef_range / hp_dev_drop_coeffs follow the context's parity
 * level: at HP_PARITY_A the limb-sharded pipeline returns canonical residues like the single-GPU entry points. */
int hp_dev_drop_apply_range_strict(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, uint64_t plain_modulus,
                                   size_t P2, size_t k0, size_t k1, const uint64_t *d_x, const uint64_t *d_clast,
                                   const uint64_t *d_addend, size_t add_poly_stride, size_t add_ct_stride, unsigned add_mask,
                                   uint64_t *d_out);

/* ---- node: several GPUs behind one handle (hehub_amd/csrc/hp_node.cpp) ------------------------------------------
 * hehub is a single-threaded CPU library; an application that holds a BATCH of ciphertexts uses all GPUs of a node through
 * this layer, from C, without Python.  A node owns one hp_ctx (own stream) and one worker thread per rank; ranks may share
 * a device (devices = {0, 0}: how the one-GPU tests exercise it).  Calls on a node are serialised.
 *
 * Batch-sharded mode (the throughput mode): item i of the batch belongs to the rank whose contiguous slice
 * [lo, hi) holds i (hp_node_slice; sizes differ by at most one); every rank runs the single-GPU entry point on its
 * slice; keys are replicated once (hp_node_replicate); there is NO exchange on the data path. */
typedef struct hp_node hp_node;
int hp_node_create(const int *devices, size_t count, hp_node **out);
void hp_node_destroy(hp_node *node);
size_t hp_node_size(const hp_node *node);
hp_ctx *hp_node_ctx(hp_node *node, size_t rank);      /* the rank's engine context, for the hp_dev_* entry points */
/* hp_ctx_set_parity_level on every rank's context: the batch-sharded entry points below then return canonical residues (level A);
 * the limb-sharded mode follows it too since round 5 (its stages run on rows the engine wrote itself: hp_dev_*_range_strict) */
int hp_node_set_parity_level(hp_node *node, int level);
const char *hp_node_last_error(hp_node *node);   /* "rank r: <message of that rank's failing call>"; the first rank with a failure of its own */
/* matrix[a * size + b] = 1 when rank a can write rank b's device memory directly (same device, or hipDeviceEnablePeerAccess
 * succeeded at hp_node_create), else 0.  The limb-sharded plan below writes peers directly where it can and stages through
 * page-locked host memory where it cannot (sender: one device-to-host copy of its block; receiver: host-to-device after the
 * rendezvous).  HP_NODE_NO_PEER in the environment at hp_node_create forces 0 for every pair a != b (tests / debugging). */
int hp_node_peer_matrix(const hp_node *node, int *matrix);
/* NUMA placement: every rank's worker thread is bound to the CPUs of the socket its GPU hangs off (PCI address of the HIP device ->
 * sysfs numa_node -> that node's cpulist, intersected with the process's affinity mask; HP_NODE_NO_AFFINITY in the environment at
 * hp_node_create leaves the threads alone).  hp_node_placement reports what was done for a rank (numa_node -1 / cpus_bound 0: the
 * platform does not say, nothing was bound); hp_device_numa is the lookup itself, for a process that places its own ranks
 * (bench.py binds each rank's process the same way). */
int hp_node_placement(const hp_node *node, size_t rank, int *numa_node, int *cpus_bound);
int hp_device_numa(int device, int *numa_node, char *cpulist, size_t cap);
int hp_node_slice(const hp_node *node, size_t total, size_t rank, size_t *lo, size_t *hi);
int hp_node_sync(hp_node *node);
/* copy a read-only host object (a key-switching key) to every rank: d_copies[rank] receives the device pointers */
int hp_node_replicate(hp_node *node, const uint64_t *h_words, size_t words, uint64_t **d_copies);
int hp_node_free_replicas(hp_node *node, uint64_t **d_copies);
/* host-resident batches (what a hehub application holds): each rank stages its slice in, computes, stages it out
 * ckks.h:270 mult + ckks.h:313 rescale_inplace / bgv mult_low_level + relinearize + mod_switch_inplace:
 * h_ct1, h_ct2 u64[batch][2][L][N] -> h_out u64[batch][2][L-1][N]; d_key[rank] from hp_node_replicate */
int hp_node_ckks_mult_relin_rescale(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, size_t batch,
                                    const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key, uint64_t *h_out);
int hp_node_bgv_mult_relin_modswitch(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus,
                                     size_t batch, const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key,
                                     uint64_t *h_out);
/* ntt.h:41-51 / :72-92 on a host-resident batch u64[batch][L][N], in place */
int hp_node_ntt(hp_node *node, size_t logn, size_t L, const uint64_t *moduli, size_t batch, uint64_t *h_x, int inverse,
                int strict);
/* device-resident slices: rank r works on counts[r] items behind d_*[r] (allocated on hp_node_ctx(node, r)) */
int hp_node_dev_ckks_mult_relin_rescale(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, const size_t *counts,
                                        const uint64_t *const *d_ct1, const uint64_t *const *d_ct2, uint64_t *const *d_key,
                                        uint64_t *const *d_out);
int hp_node_dev_bgv_mult_relin_modswitch(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus,
                                         const size_t *counts, const uint64_t *const *d_ct1, const uint64_t *const *d_ct2,
                                         uint64_t *const *d_key, uint64_t *const *d_out);
/* Limb-sharded ("latency") mode: ONE batch processed by all ranks, cut by output modulus (the limb-range stages above);
 * rank r owns a contiguous range of q_0..q_{L-1}, p (hp_node_sharded_range; sizes differ by at most one, the special prime
 * in a smallest range).  Exchanges are direct peer writes: the owner copies its limbs into every peer's buffer (one xGMI link
 * per shard, no ring, no padding), ordered by HIP events; all buffers belong to the plan.  Pairs without peer access
 * (hp_node_peer_matrix) go through the owner's page-locked staging buffer instead.  With L+1 moduli over W ranks the
 * speed-up is bounded by (L+1) / ceil((L+1)/W): 11 moduli over 8 GPUs -> 5.5 x.  plain_modulus 0: CKKS pipeline. */
typedef struct hp_node_sharded hp_node_sharded;
/* How the plans made FROM NOW ON exchange their limbs (the four exchanges of one multiplication: the all-gather of the key-switch
 * coefficient limbs, the broadcasts of the two dropped limbs' coefficients, the all-gather of the result limbs):
 *   HP_TRANSPORT_PEER    (default) direct peer writes, one xGMI link per shard, ordered by HIP events (staged through page-locked host
 *                        memory for a pair without peer access)
 *   HP_TRANSPORT_RCCL    the collective the north star names: ncclAllGather of the packed, padded per-rank parts and ncclBroadcast from
 *                        the owner, on the ranks' streams.  librccl.so is loaded when this is first asked for (the engine does not link
 *                        it); one communicator rank per node rank (ncclCommInitAll over the node's devices), so every rank needs its
 *                        own device -- HP_EUNSUPPORTED otherwise, or when the library cannot be loaded
 *   HP_TRANSPORT_PACKED  the same packed send / receive buffers as _RCCL, moved by plain device copies (ranks that share a GPU cannot
 *                        form a communicator: this is how the one-GPU tests cover the packing)
 * Every sum over the digits for an output modulus is formed on ONE rank in hehub's order (rgsw.cpp:98-153) whatever the transport:
 * the words are identical under all three (tests/test_gpu_node.py). */
enum { HP_TRANSPORT_PEER = 0, HP_TRANSPORT_RCCL = 1, HP_TRANSPORT_PACKED = 2 };
int hp_node_set_transport(hp_node *node, int transport);
int hp_node_get_transport(const hp_node *node);
int hp_node_sharded_create(hp_node *node, size_t logn, size_t L, const uint64_t *moduli_ext, uint64_t plain_modulus, size_t batch,
                           hp_node_sharded **out);
void hp_node_sharded_destroy(hp_node_sharded *plan);
int hp_node_sharded_range(const hp_node_sharded *plan, size_t rank, size_t *k0, size_t *k1);
/* host operands (copied to every rank), result gathered on rank 0 and copied back */
int hp_node_sharded_mult(hp_node_sharded *plan, const uint64_t *h_ct1, const uint64_t *h_ct2, uint64_t *const *d_key,
                         uint64_t *h_out);
/* operands already replicated on every rank; every rank ends up with the whole result in d_out[rank] */
int hp_node_sharded_mult_dev(hp_node_sharded *plan, const uint64_t *const *d_ct1, const uint64_t *const *d_ct2,
                             uint64_t *const *d_key, uint64_t *const *d_out);

/* ---- extensions beyond the reference ---------------------------------------------------
 * hehub throws for both cases; these entry points are additions, not replacements, and are pinned by equivalence to
 * the reference-parity entry points (tests/test_extensions.py):
 * (1) a key-switching key generated for key_L0 ciphertext moduli used at a LOWER level L <= key_L0 (hehub requires
 *     exactly L+1 limbs, rgsw.cpp:84-87, so one key serves one level only): key u64[key_L0][2][key_L0+1][N]; digit rows
 *     j >= L and modulus columns L..key_L0-1 are ignored, the special prime is the last column.  moduli_ext stays
 *     q_0..q_{L-1}, p.  Identical, word for word, to calling the plain entry point with the extracted sub-key.
 * (2) rescale by several primes (hehub: "under development", rescaling.cpp:83-85) = successive exact one-prime drops. */
int hp_dev_ext_prod_montgomery_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                                  size_t batch, const uint64_t *d_pt, const uint64_t *d_key, uint64_t *d_out);
int hp_dev_ckks_relinearize_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                               size_t batch, const uint64_t *d_quad, const uint64_t *d_key, uint64_t *d_out);
int hp_dev_ckks_rotate_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                          size_t step, const uint64_t *d_ct, const uint64_t *d_rot_key, uint64_t *d_out);
int hp_dev_ckks_conjugate_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                             size_t batch, const uint64_t *d_ct, const uint64_t *d_conj_key, uint64_t *d_out);
/* (3) ckks.h:284 rotate / ckks.h:282 conjugate of `batch` ciphertexts, EACH WITH ITS OWN KEY AND STEP: hehub's circuits rotate one
 *     vector by many steps, every step under its own key (src/circuits/linear_algebra.h:123-130: rotate(ct_vec, rot_key_set[s]) for
 *     2 (width - 1) values of s) -- independent calls, but no two share a key, so the one-key batch form above does not apply.
 *     d_ct u64[batch][2][L][N]; steps[b] and d_keys[b] (a HOST array of device addresses, key b laid out as for
 *     hp_dev_ckks_rotate_at with key_L0) belong to ciphertext b; conj (may be NULL) marks the ciphertexts that are conjugated
 *     instead (their step is ignored).  -> d_out u64[batch][2][L][N], word for word what `batch` calls of
 *     hp_dev_ckks_rotate_at / hp_dev_ckks_conjugate_at with batch 1 write (ckks/arith.cpp:75-93). */
int hp_dev_ckks_rotate_many(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                            const size_t *steps, const unsigned char *conj, const uint64_t *d_ct,
                            const uint64_t *const *d_keys, uint64_t *d_out);
/*     The same with the ciphertexts' polynomials anywhere in device memory: d_polys[2 b + h] (a HOST array of device addresses) is
 *     polynomial h of ciphertext b, u64[L][N] -- an application's ciphertexts are separate objects, and the rotations of ONE vector
 *     (linear_algebra.h:123-130) all read the same two polynomials: no packed copy of the batch is made. */
int hp_dev_ckks_rotate_many_rows(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext, size_t batch,
                                 const size_t *steps, const unsigned char *conj, const uint64_t *const *d_polys,
                                 const uint64_t *const *d_keys, uint64_t *d_out);
/* (4) The fused multiplication pipelines with the OPERANDS BY ADDRESS: d_polys[4 b + {0, 1, 2, 3}] (a HOST array of device addresses)
 *     = polynomials ct1[b][0], ct1[b][1], ct2[b][0], ct2[b][1], u64[L][N] each, anywhere in device memory.  An application's
 *     ciphertexts are separate objects (hehub's are: ckks.h:73-93); the packed forms above would cost a gather of 4 L limbs per pair
 *     (12 % of a C3 step), here the tensor product -- the only kernel that reads the operands -- takes the addresses as arguments.
 *     Same words as hp_dev_ckks_mult_relin_rescale_at / hp_dev_bgv_mult_relin_modswitch. */
int hp_dev_ckks_mult_relin_rescale_rows(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                                        size_t batch, const uint64_t *const *d_polys, const uint64_t *d_key, uint64_t *d_out);
int hp_dev_bgv_mult_relin_modswitch_rows(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli_ext,
                                         uint64_t plain_modulus, size_t batch, const uint64_t *const *d_polys,
                                         const uint64_t *d_key, uint64_t *d_out);
int hp_dev_ckks_mult_relin_rescale_at(hp_ctx *ctx, size_t logn, size_t L, size_t key_L0, const uint64_t *moduli_ext,
                                      size_t batch, const uint64_t *d_ct1, const uint64_t *d_ct2,
                                      const uint64_t *d_key, uint64_t *d_out);
/* (3) many -> many base conversion (hehub: "under development", rns_transform.cpp:123): the exact CRT value x of every
 *     coefficient, centred around Q/2 the way the many -> one CRT branch does it (x mod m below the half,
 *     m - ((Q - x) mod m) from the half on), in every new modulus.  in u64[batch][L][N] -> out u64[batch][Lnew][N] */
int hp_dev_rns_base_many_to_many(hp_ctx *ctx, size_t n, size_t L, const uint64_t *old_moduli, size_t Lnew,
                                 const uint64_t *new_moduli, size_t batch, const uint64_t *d_in, uint64_t *d_out);
/* (4) hybrid key switch: alpha consecutive ciphertext moduli form one digit (dnum = ceil(L/alpha) digits) and k special
 *     primes p_0..p_{k-1} (product >= every digit's modulus product) replace hehub's single one: dnum*(L+k) - L digit
 *     transforms per switch instead of L*L.  moduli_ext = q_0..q_{L-1}, p_0..p_{k-1}.  The key has its own format,
 *     u64[dnum][2][L+k][N], NTT + Montgomery form like hehub's: row d is an RLWE encryption, under all L+k moduli, of
 *     (P mod q_i) * s_from in the limbs i of digit d and of 0 elsewhere -- hehub's key is the case alpha = 1, k = 1.
 *     Every step is exact integer arithmetic (ModUp / ModDown by mixed-radix composition), pinned by an exact integer
 *     model and by decryption (tests/test_hks.py); results are NOT comparable with hehub's (different keys, less noise).
 *     Limits: alpha <= 8, at most 16 digits, k <= 16 (one fused conversion kernel up to k = 8), L + k <= 32.
 *     hp_dev_hks_switch: pt u64[batch][L][N] (NTT form) -> out u64[batch][2][L][N] with out0 + out1*s ~ pt*s_from. */
int hp_dev_hks_switch(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext, size_t batch,
                      const uint64_t *d_pt, const uint64_t *d_key, uint64_t *d_out);
/* ckks::rotate / conjugate with a hybrid rotation / conjugation key: ct u64[batch][2][L][N] -> out u64[batch][2][L][N] */
int hp_dev_ckks_rotate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext,
                           size_t batch, size_t step, const uint64_t *d_ct, const uint64_t *d_rot_key, uint64_t *d_out);
int hp_dev_ckks_conjugate_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha, const uint64_t *moduli_ext,
                              size_t batch, const uint64_t *d_ct, const uint64_t *d_conj_key, uint64_t *d_out);
/* ckks::mult_low_level + relinearisation with a hybrid key + rescale by q_{L-1}: out u64[batch][2][L-1][N].
 * For N = 2^11 .. 2^15 ModDown and the rescale share one transform per limb: the residues of hp_dev_hks_switch followed by
 * hp_dev_ckks_rescale, in a lazy representative (< 2q) of their own; HP_HKS_TWO_STEP=1 in the environment at hp_ctx_create
 * runs the two steps separately and reproduces those words. */
int hp_dev_ckks_mult_relin_rescale_hks(hp_ctx *ctx, size_t logn, size_t L, size_t k, size_t alpha,
                                       const uint64_t *moduli_ext, size_t batch, const uint64_t *d_ct1,
                                       const uint64_t *d_ct2, const uint64_t *d_key, uint64_t *d_out);
/* ct u64[batch][2][L][N] -> out u64[batch][2][L-drops][N]; d_tmp: 2 * batch*2*(L-1)*N words (may be NULL for drops == 1) */
int hp_dev_ckks_rescale_n(hp_ctx *ctx, size_t logn, size_t L, const uint64_t *moduli, size_t drops, size_t batch,
                          const uint64_t *d_ct, uint64_t *d_tmp, uint64_t *d_out);

/* ---- wire / on-disk format (hehub itself has none) -------------------------------------
 * "HEHUBAMD" header + moduli + the words in device order + FNV-1a-64 trailer; the exact byte layout is documented in
 * hehub_amd/csrc/hp_wire.cpp.  Loading a key or ciphertext is one validation pass and one host-to-device copy. */
typedef enum { HP_WIRE_POLY = 1, HP_WIRE_CT = 2, HP_WIRE_QUAD_CT = 3, HP_WIRE_KSK = 4 } hp_wire_kind;
typedef struct {
    uint32_t kind;           /* hp_wire_kind */
    uint32_t log_dimension;  /* log2 N */
    uint32_t limbs;          /* limbs per polynomial (a key: L+1) */
    uint32_t polys;          /* 1, 2, 3; a key: 2*digits, rgsw[j][half] digit-major */
    uint32_t rep_form;       /* 0 coefficient, 1 NTT value */
    uint64_t scheme_scalar;  /* IEEE-754 bits of the CKKS scaling factor / BGV plain modulus / 0 */
} hp_wire_desc;
/* the format's checksum: FNV-1a-64 over `bytes` bytes (little-endian words as they lie in memory) */
uint64_t hp_wire_fnv1a64(const void *data, size_t bytes);
size_t hp_wire_payload_words(const hp_wire_desc *d);   /* 0 for an invalid descriptor */
size_t hp_wire_bytes(const hp_wire_desc *d);
/* host words -> buffer, and back (payload_offset: where the words start inside buf) */
int hp_wire_pack(const hp_wire_desc *d, const uint64_t *moduli, const uint64_t *words, void *buf, size_t cap);
int hp_wire_unpack(const void *buf, size_t len, hp_wire_desc *d, uint64_t *moduli, size_t moduli_cap,
                   size_t *payload_offset);
/* buffer -> device words (validated: magic, version, sizes, checksum), device words -> buffer */
int hp_dev_wire_load(hp_ctx *ctx, const void *buf, size_t len, uint64_t *d_words);
int hp_dev_wire_store(hp_ctx *ctx, const hp_wire_desc *d, const uint64_t *moduli, const uint64_t *d_words, void *buf,
                      size_t cap);

/* ---- in-library kernel timing (HIP events on the ctx stream) -------------- */
/* Between hp_prof_begin and hp_prof_end every kernel launch of the named
 * family is bracketed by hipEvents on the stream it is launched on.
 * hp_prof_end synchronises and returns launches and total milliseconds. */
int hp_prof_begin(hp_ctx *ctx, const char *kernel_family);
int hp_prof_end(hp_ctx *ctx, size_t *launches, double *total_ms);
/* With the family "*" every launch of every family is bracketed; this returns the totals per family (at most `cap` of them, in
 * order of first appearance): names[i] is a static string ("tensor", "intt", "ntt", "ks_inner", "ntt_drop", "elem", "copy", ...). */
int hp_prof_end_families(hp_ctx *ctx, size_t cap, const char **names, size_t *launches, double *total_ms, size_t *count);

#ifdef __cplusplus
}
#endif
#endif
