/*
 * hehub_oracle.c -- CPU ORACLE (test infrastructure, see hehub_oracle.h).
 *
 * Plain C restatement of the integer formulas primihub/hehub applies on the
 * NTT / mod-arith / key-switch / rescale path.  Each function names the
 * reference lines whose arithmetic it restates.  The restatement keeps the
 * reference's *per-element sequence of wrapping-u64 / u128 operations* so raw
 * lazy words (not only residues) agree; it does not keep the reference's
 * containers, caches or loop nests.
 *
 * Parity status: PINNED against oracle/_ref (the compiled reference) and
 * tests/golden/ -- see the header.
 */
#include "hehub_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef orc_u64 u64;
typedef unsigned __int128 u128;
typedef __int128 i128;

/* ===================================================================== */
/* scalar helpers                                                        */
/* ===================================================================== */

/* floor(b * 2^64 / q): the "Harvey" companion word (ntt.cpp:56, rns.cpp:146) */
u64 orc_harvey_quotient(u64 b, u64 q) { return (u64)(((u128)b << 64) / q); }

/* mod_arith.h:74-78 */
u64 orc_mul_mod_harvey_lazy(u64 q, u64 a, u64 b, u64 bh) {
    u64 qhat = (u64)(((u128)a * bh) >> 64);
    return (u64)((u128)a * b - (u128)qhat * q);
}

/* mod_arith.cpp:19-47,136-149: Bezout coefficient of elem in
 * prime*x + elem*y = 1, lifted into [0, prime). */
u64 orc_inverse_mod_prime(u64 elem, u64 prime) {
    i128 r0 = (i128)prime, r1 = (i128)elem;
    i128 y0 = 0, y1 = 1;
    while (r1 != 0) {
        i128 quo = r0 / r1;
        i128 r2 = r0 - quo * r1;
        i128 y2 = y0 - quo * y1;
        r0 = r1; r1 = r2;
        y0 = y1; y1 = y2;
    }
    if (y0 < 0) y0 += (i128)prime;
    return (u64)y0;
}

/* ntt.cpp:9-24 (left-to-right square and multiply; result is canonical) */
u64 orc_pow_mod(u64 q, u64 base, u64 index) {
    u64 power = 1;
    int top = -1;
    for (int b = 63; b >= 0; b--) {
        if ((index >> b) & 1) { top = b; break; }
    }
    for (int b = top; b >= 0; b--) {
        power = (u64)((u128)power * power % q);
        if ((index >> b) & 1) power = (u64)((u128)power * base % q);
    }
    return power;
}

/* ntt.cpp:26-39 */
int orc_get_2nth_unity_root(u64 q, u64 n, u64 *root) {
    if ((q - 1) % (2 * n) != 0) return -1;
    u64 g = 2;
    while (orc_pow_mod(q, g, (q - 1) / 2) != q - 1) g++;
    *root = orc_pow_mod(q, g, (q - 1) / (2 * n));
    return 0;
}

/* permutation.h:41-55 (bit_len <= 16) */
u64 orc_bit_rev(u64 x, int bit_len) {
    u64 r = 0;
    for (int i = 0; i < bit_len; i++) r |= ((x >> i) & 1) << (bit_len - 1 - i);
    return r;
}

/* mod_arith.cpp:49-52: -q^{-1} mod 2^64 (Newton iteration gives the same
 * unique value the reference obtains through xgcd). */
u64 orc_minus_q_inv_mod_2to64(u64 q) {
    u64 inv = q; /* correct to 3 bits for odd q */
    for (int i = 0; i < 6; i++) inv *= 2 - q * inv;
    return (u64)0 - inv;
}

/* mod_arith.cpp:54-57: ((2^64-1) mod q) + 1  (== q when q | 2^64, never for odd q>1) */
u64 orc_2to64_mod_q(u64 q) { return (~(u64)0) % q + 1; }

/* ===================================================================== */
/* batched modular kernels                                               */
/* ===================================================================== */

/* mod_arith.cpp:9-17 */
void orc_batched_barrett_lazy(u64 q, size_t n, u64 *v) {
    u64 c = (~(u64)0) / q;
    for (size_t i = 0; i < n; i++) {
        u64 qhat = (u64)(((u128)v[i] * c) >> 64);
        v[i] -= q * qhat;
    }
}

/* mod_arith.h:58-63 */
void orc_batched_reduce_strict(u64 q, size_t n, u64 *v) {
    for (size_t i = 0; i < n; i++) v[i] -= (v[i] >= q) ? q : 0;
}

/* mod_arith.h:18-25 */
void orc_batched_barrett(u64 q, size_t n, u64 *v) {
    orc_batched_barrett_lazy(q, n, v);
    orc_batched_reduce_strict(q, n, v);
}

/* mod_arith.cpp:64-92: Montgomery reduction of a*b followed by a Harvey
 * multiplication by 2^64 mod q. */
void orc_batched_mul_mod_hybrid_lazy(u64 q, size_t n, const u64 *a, const u64 *b,
                                     u64 *out) {
    const u64 m = orc_minus_q_inv_mod_2to64(q);
    const u64 r = orc_2to64_mod_q(q);
    const u64 rh = orc_harvey_quotient(r, q);
    for (size_t i = 0; i < n; i++) {
        u128 prod = (u128)a[i] * b[i];
        u64 u = (u64)prod * m;
        u64 t = (u64)((prod + (u128)u * q) >> 64);
        u64 t2 = (u64)(((u128)t * rh) >> 64);
        out[i] = (u64)((u128)t * r - (u128)t2 * q);
    }
}

/* mod_arith.cpp:94-111 */
void orc_batched_mul_mod_barrett_lazy(u64 q, size_t n, const u64 *a, const u64 *b,
                                      u64 *out) {
    u128 c = (~(u128)0) / q;
    u64 ch = (u64)(c >> 64), cl = (u64)c;
    for (size_t i = 0; i < n; i++) {
        u128 prod = (u128)a[i] * b[i];
        u64 ah = (u64)(prod >> 64), al = (u64)prod;
        u64 qhat = ah * ch + (u64)((((u128)ah * cl) + ((u128)al * ch)) >> 64);
        out[i] = (u64)(prod - (u128)q * qhat);
    }
}

/* mod_arith.cpp:113-134 */
void orc_batched_montgomery_128_lazy(u64 q, size_t n, const u64 *in128, u64 *out) {
    const u64 m = orc_minus_q_inv_mod_2to64(q);
    for (size_t i = 0; i < n; i++) {
        u128 a = ((u128)in128[2 * i + 1] << 64) | in128[2 * i];
        u64 u = (u64)a * m;
        out[i] = (u64)((a + (u128)u * q) >> 64);
    }
}

/* ===================================================================== */
/* twiddle tables                                                        */
/* ===================================================================== */

static int log_modulus_of(u64 q) { return (int)(u64)(log2((double)q) + 0.5); }

/* ntt.cpp:41-58 */
int orc_ntt_factors(u64 q, size_t logn, u64 *seq, u64 *seq_harvey) {
    if (log_modulus_of(q) > 59) return -2;
    const size_t n = (size_t)1 << logn;
    u64 psi;
    if (orc_get_2nth_unity_root(q, n, &psi) != 0) return -1;
    for (size_t i = 0; i < n; i++) {
        seq[i] = orc_pow_mod(q, psi, orc_bit_rev(i, (int)logn));
        seq_harvey[i] = orc_harvey_quotient(seq[i], q);
    }
    return 0;
}

/* ntt.cpp:59-90.  Entry N-1 is never written by the reference either (its
 * vector is zero-initialised), entries N..2N-1 hold psi^-i * N^-1. */
int orc_intt_factors(u64 q, size_t logn, u64 *seq, u64 *seq_harvey) {
    if (log_modulus_of(q) > 59) return -2;
    const size_t n = (size_t)1 << logn;
    u64 psi;
    if (orc_get_2nth_unity_root(q, n, &psi) != 0) return -1;
    memset(seq, 0, 2 * n * sizeof(u64));
    memset(seq_harvey, 0, 2 * n * sizeof(u64));
    const u64 psi_inv = orc_pow_mod(q, psi, 2 * n - 1);
    for (size_t l = 0; l < logn; l++) {
        const size_t start = ((size_t)1 << l) - 1;
        const u64 stride = (u64)1 << (logn - l);
        for (size_t i = 0; i < ((size_t)1 << l); i++) {
            seq[start + i] = orc_pow_mod(q, psi_inv, orc_bit_rev(i, (int)l) * stride);
            seq_harvey[start + i] = orc_harvey_quotient(seq[start + i], q);
        }
    }
    const u64 n_inv = q - ((q - 1) >> logn);
    const u64 n_inv_h = orc_harvey_quotient(n_inv, q);
    for (size_t i = 0; i < n; i++) {
        u64 w = orc_mul_mod_harvey_lazy(q, orc_pow_mod(q, psi_inv, i), n_inv, n_inv_h);
        w -= (w >= q) ? q : 0;
        seq[n + i] = w;
        seq_harvey[n + i] = orc_harvey_quotient(w, q);
    }
    return 0;
}

/* explicit table cache (the reference uses process-global std::maps,
 * ntt.cpp:107-143) */
typedef struct {
    u64 q;
    size_t logn;
    int inverse;
    u64 *seq, *seq_harvey;
} table_t;
static table_t *g_tables = NULL;
static size_t g_ntables = 0, g_captables = 0;

void orc_clear_cache(void) {
    for (size_t i = 0; i < g_ntables; i++) { free(g_tables[i].seq); free(g_tables[i].seq_harvey); }
    free(g_tables);
    g_tables = NULL; g_ntables = g_captables = 0;
}

static int get_table(u64 q, size_t logn, int inverse, const table_t **out) {
    for (size_t i = 0; i < g_ntables; i++) {
        if (g_tables[i].q == q && g_tables[i].logn == logn && g_tables[i].inverse == inverse) {
            *out = &g_tables[i];
            return 0;
        }
    }
    const size_t n = (size_t)1 << logn;
    const size_t len = inverse ? 2 * n : n;
    u64 *seq = (u64 *)malloc(len * sizeof(u64));
    u64 *sh = (u64 *)malloc(len * sizeof(u64));
    int rc = inverse ? orc_intt_factors(q, logn, seq, sh) : orc_ntt_factors(q, logn, seq, sh);
    if (rc != 0) { free(seq); free(sh); return rc; }
    if (g_ntables == g_captables) {
        g_captables = g_captables ? 2 * g_captables : 16;
        g_tables = (table_t *)realloc(g_tables, g_captables * sizeof(table_t));
    }
    g_tables[g_ntables] = (table_t){q, logn, inverse, seq, sh};
    *out = &g_tables[g_ntables++];
    return 0;
}

/* ===================================================================== */
/* transforms                                                            */
/* ===================================================================== */

/* the radix-2 lazy butterfly of ntt.cpp:160-166 / :199-205 */
static inline void butterfly(u64 q, u64 two_q, u64 *lo, u64 *hi, u64 w, u64 wh) {
    u64 t = orc_mul_mod_harvey_lazy(q, *hi, w, wh);
    *hi = *lo + two_q - t;
    *lo = *lo + t;
}

/* final fold of ntt.cpp:171-175 / :214-218 */
static inline u64 shift_fold(u64 x, u64 q, int k, u64 fix) { return x - ((x >> k) - fix) * q; }

/* ntt.cpp:145-176 */
int orc_ntt_negacyclic_inplace_lazy(size_t logn, u64 q, u64 *x) {
    const table_t *tb;
    int rc = get_table(q, logn, 0, &tb);
    if (rc != 0) return rc;
    const size_t n = (size_t)1 << logn;
    const u64 two_q = 2 * q;
    size_t idx = 1;
    for (size_t span = n; span >= 2; span >>= 1) {
        const size_t gap = span >> 1;
        for (size_t start = 0; start < n; start += span, idx++) {
            const u64 w = tb->seq[idx], wh = tb->seq_harvey[idx];
            for (size_t l = start; l < start + gap; l++) butterfly(q, two_q, &x[l], &x[l + gap], w, wh);
        }
    }
    const int k = log_modulus_of(q);
    const u64 fix = (q >= ((u64)1 << k)) ? 1 : 0;
    for (size_t i = 0; i < n; i++) x[i] = shift_fold(x[i], q, k, fix);
    return 0;
}

/* ntt.cpp:178-223.  The reference gathers through the bit-reversal
 * permutation, runs the forward-shaped network and gathers back; viewed in the
 * caller's index space that is a network whose stage s (s = 0..logN-1) pairs
 * i and i + 2^s and uses the level-s twiddle of (i mod 2^s).  Both forms do
 * the same butterflies on the same words; the reference form is kept. */
int orc_intt_negacyclic_inplace_lazy(size_t logn, u64 q, u64 *x) {
    const table_t *tb;
    int rc = get_table(q, logn, 1, &tb);
    if (rc != 0) return rc;
    const size_t n = (size_t)1 << logn;
    const u64 two_q = 2 * q;
    u64 *y = (u64 *)malloc(n * sizeof(u64));
    for (size_t i = 0; i < n; i++) y[i] = x[orc_bit_rev(i, (int)logn)];
    size_t idx = 0;
    for (size_t span = n; span >= 2; span >>= 1) {
        const size_t gap = span >> 1;
        for (size_t start = 0; start < n; start += span, idx++) {
            const u64 w = tb->seq[idx], wh = tb->seq_harvey[idx];
            for (size_t l = start; l < start + gap; l++) butterfly(q, two_q, &y[l], &y[l + gap], w, wh);
        }
    }
    const int k = log_modulus_of(q);
    const u64 fix = (q >= ((u64)1 << k)) ? 1 : 0;
    for (size_t i = 0; i < n; i++) {
        u64 v = shift_fold(y[orc_bit_rev(i, (int)logn)], q, k, fix);
        x[i] = orc_mul_mod_harvey_lazy(q, v, tb->seq[n + i], tb->seq_harvey[n + i]);
    }
    free(y);
    return 0;
}

/* ===================================================================== */
/* RnsIntVec operators                                                   */
/* ===================================================================== */

/* rns.cpp:58-87 */
void orc_poly_add_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, const u64 *b) {
    for (size_t k = 0; k < L; k++) {
        const u64 two_q = 2 * moduli[k];
        u64 *s = self + k * n;
        const u64 *o = b + k * n;
        for (size_t i = 0; i < n; i++) {
            u64 v = s[i] + o[i];
            s[i] = v - ((v >= two_q) ? two_q : 0);
        }
    }
}

/* rns.cpp:89-118 */
void orc_poly_sub_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, const u64 *b) {
    for (size_t k = 0; k < L; k++) {
        const u64 two_q = 2 * moduli[k];
        u64 *s = self + k * n;
        const u64 *o = b + k * n;
        for (size_t i = 0; i < n; i++) {
            u64 v = s[i] + (two_q - o[i]);
            s[i] = v - ((v >= two_q) ? two_q : 0);
        }
    }
}

/* rns.cpp:120-140 */
void orc_poly_mul(size_t n, size_t L, const u64 *moduli, const u64 *a, const u64 *b, u64 *out) {
    for (size_t k = 0; k < L; k++)
        orc_batched_mul_mod_hybrid_lazy(moduli[k], n, a + k * n, b + k * n, out + k * n);
}

static void limb_scalar_mul(u64 q, size_t n, u64 *x, u64 scalar) {
    const u64 s = scalar % q;
    const u64 sh = orc_harvey_quotient(s, q);
    for (size_t i = 0; i < n; i++) x[i] = orc_mul_mod_harvey_lazy(q, x[i], s, sh);
}

/* rns.cpp:142-153 */
void orc_poly_scalar_mul_inplace(size_t n, size_t L, const u64 *moduli, u64 *self, u64 scalar) {
    for (size_t k = 0; k < L; k++) limb_scalar_mul(moduli[k], n, self + k * n, scalar);
}

/* rns.cpp:155-171 */
void orc_poly_rns_scalar_mul_inplace(size_t n, size_t L, const u64 *moduli, u64 *self,
                                     const u64 *rns_scalar) {
    for (size_t k = 0; k < L; k++) limb_scalar_mul(moduli[k], n, self + k * n, rns_scalar[k]);
}

/* ntt.h:41-51 */
int orc_poly_ntt(size_t logn, size_t L, const u64 *moduli, u64 *x) {
    const size_t n = (size_t)1 << logn;
    for (size_t k = 0; k < L; k++) {
        int rc = orc_ntt_negacyclic_inplace_lazy(logn, moduli[k], x + k * n);
        if (rc != 0) return rc;
    }
    return 0;
}

/* ntt.h:72-82 */
int orc_poly_intt(size_t logn, size_t L, const u64 *moduli, u64 *x) {
    const size_t n = (size_t)1 << logn;
    for (size_t k = 0; k < L; k++) {
        int rc = orc_intt_negacyclic_inplace_lazy(logn, moduli[k], x + k * n);
        if (rc != 0) return rc;
    }
    return 0;
}

/* mod_arith.h:65-72 */
void orc_poly_reduce_strict(size_t n, size_t L, const u64 *moduli, u64 *x) {
    for (size_t k = 0; k < L; k++) orc_batched_reduce_strict(moduli[k], n, x + k * n);
}

/* ===================================================================== */
/* automorphisms                                                         */
/* ===================================================================== */

/* permutation.cpp:59-75 */
void orc_poly_involution(size_t logn, size_t L, const u64 *in, u64 *out) {
    const size_t n = (size_t)1 << logn;
    for (size_t k = 0; k < L; k++)
        for (size_t i = 0; i < n; i++) out[k * n + i] = in[k * n + (n - 1 - i)];
}

/* permutation.cpp:28-57: slot rotation by `step` = substitution X -> X^(3^step) */
void orc_poly_cycle(size_t logn, size_t L, size_t step, const u64 *in, u64 *out) {
    const size_t n = (size_t)1 << logn;
    const uint32_t mask = (uint32_t)((1u << (logn + 1)) - 1);
    uint32_t factor = 1;
    for (size_t s = 0; s < step; s++) factor *= 3u;
    factor &= mask;
    uint32_t pw = 1; /* 3^i mod 2^32 */
    for (size_t i = 0; i < n / 2; i++, pw *= 3u) {
        const uint32_t old_idx = pw & mask;
        const size_t from = (size_t)orc_bit_rev((old_idx - 1) / 2, (int)logn);
        const uint32_t new_idx = (old_idx * factor) & mask;
        const size_t to = (size_t)orc_bit_rev((new_idx - 1) / 2, (int)logn);
        for (size_t k = 0; k < L; k++) {
            out[k * n + to] = in[k * n + from];
            out[k * n + (n - 1 - to)] = in[k * n + (n - 1 - from)];
        }
    }
}

/* ===================================================================== */
/* key switch                                                            */
/* ===================================================================== */

/* rgsw.cpp:57-156 */
int orc_ext_prod_montgomery(size_t logn, size_t L, const u64 *moduli_ext, const u64 *pt,
                            const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    const size_t Le = L + 1;
    int rc = 0;
    /* (i) coefficient form of every limb, strictly reduced (rgsw.cpp:103-105) */
    u64 *coef = (u64 *)malloc(L * n * sizeof(u64));
    memcpy(coef, pt, L * n * sizeof(u64));
    rc = orc_poly_intt(logn, L, moduli_ext, coef);
    if (rc == 0) orc_poly_reduce_strict(n, L, moduli_ext, coef);
    /* (ii) digit matrix D[j][k] (rgsw.cpp:98-119) */
    u64 *dig = (u64 *)malloc(L * Le * n * sizeof(u64));
    for (size_t j = 0; j < L && rc == 0; j++) {
        for (size_t k = 0; k < Le && rc == 0; k++) {
            u64 *d = dig + (j * Le + k) * n;
            if (k == j) {
                memcpy(d, pt + j * n, n * sizeof(u64));
            } else {
                memcpy(d, coef + j * n, n * sizeof(u64));
                rc = orc_ntt_negacyclic_inplace_lazy(logn, moduli_ext[k], d);
            }
        }
    }
    /* (iii) u128 inner product + Montgomery reduction (rgsw.cpp:121-153) */
    if (rc == 0) {
        u64 *acc = (u64 *)malloc(2 * n * sizeof(u64));
        for (size_t half = 0; half < 2; half++) {
            for (size_t k = 0; k < Le; k++) {
                memset(acc, 0, 2 * n * sizeof(u64));
                for (size_t j = 0; j < L; j++) {
                    const u64 *d = dig + (j * Le + k) * n;
                    const u64 *g = key + ((j * 2 + half) * Le + k) * n;
                    for (size_t i = 0; i < n; i++) {
                        u128 a = ((u128)acc[2 * i + 1] << 64) | acc[2 * i];
                        a += (u128)d[i] * g[i];
                        acc[2 * i] = (u64)a;
                        acc[2 * i + 1] = (u64)(a >> 64);
                    }
                }
                orc_batched_montgomery_128_lazy(moduli_ext[k], n, acc, out + (half * Le + k) * n);
            }
        }
        free(acc);
    }
    free(dig);
    free(coef);
    return rc;
}

/* ===================================================================== */
/* dropping the last prime                                               */
/* ===================================================================== */

/* shared skeleton of rescaling.cpp:46-75 and mod_switch.cpp:45-77 for ONE
 * polynomial u64[L][N] -> u64[L-1][N].  bgv==0: CKKS. */
static int drop_last_prime_poly(size_t logn, size_t L, const u64 *moduli, int bgv, u64 t,
                                const u64 *poly, u64 *out) {
    const size_t n = (size_t)1 << logn;
    const u64 q_last = moduli[L - 1];
    const u64 half_q_last = q_last / 2;
    int rc;
    u64 *c = (u64 *)malloc(n * sizeof(u64));
    memcpy(c, poly + (L - 1) * n, n * sizeof(u64));
    rc = orc_intt_negacyclic_inplace_lazy(logn, q_last, c);
    if (rc != 0) { free(c); return rc; }
    if (bgv) limb_scalar_mul(q_last, n, c, orc_inverse_mod_prime(t, q_last)); /* mod_switch.cpp:49 */
    orc_batched_reduce_strict(q_last, n, c);

    u64 *rem = (u64 *)malloc((L - 1) * n * sizeof(u64));
    for (size_t k = 0; k + 1 < L; k++) {
        const u64 q = moduli[k];
        const u64 r = q_last % q;
        u64 *d = rem + k * n;
        memcpy(d, c, n * sizeof(u64));
        orc_batched_barrett(q, n, d);
        for (size_t i = 0; i < n; i++)
            if (c[i] >= half_q_last) d[i] += q - r;
    }
    if (bgv) orc_poly_scalar_mul_inplace(n, L - 1, moduli, rem, t); /* mod_switch.cpp:70 */
    rc = orc_poly_ntt(logn, L - 1, moduli, rem);
    if (rc == 0) {
        memcpy(out, poly, (L - 1) * n * sizeof(u64));
        orc_poly_sub_inplace(n, L - 1, moduli, out, rem);
        for (size_t k = 0; k + 1 < L; k++)
            limb_scalar_mul(moduli[k], n, out + k * n, orc_inverse_mod_prime(q_last, moduli[k]));
        if (bgv) orc_poly_scalar_mul_inplace(n, L - 1, moduli, out, q_last % t); /* mod_switch.cpp:76 */
    }
    free(rem);
    free(c);
    return rc;
}

/* rescaling.cpp:14-78 (the scaling-factor bookkeeping is the caller's) */
int orc_ckks_rescale_by_one_prime(size_t logn, size_t L, const u64 *moduli, const u64 *ct,
                                  u64 *out) {
    if (L < 2) return -3;
    const size_t n = (size_t)1 << logn;
    for (size_t h = 0; h < 2; h++) {
        int rc = drop_last_prime_poly(logn, L, moduli, 0, 0, ct + h * L * n, out + h * (L - 1) * n);
        if (rc != 0) return rc;
    }
    return 0;
}

/* mod_switch.cpp:13-78 */
int orc_bgv_mod_drop_one_prime(size_t logn, size_t L, const u64 *moduli, u64 t, const u64 *ct,
                               u64 *out) {
    if (L < 2) return -3;
    const size_t n = (size_t)1 << logn;
    for (size_t h = 0; h < 2; h++) {
        int rc = drop_last_prime_poly(logn, L, moduli, 1, t, ct + h * L * n, out + h * (L - 1) * n);
        if (rc != 0) return rc;
    }
    return 0;
}

/* ===================================================================== */
/* scheme level                                                          */
/* ===================================================================== */

/* ckks/arith.cpp:55-62, bgv/arith.cpp:59-69 */
void orc_mult_low_level(size_t n, size_t L, const u64 *moduli, const u64 *ct1, const u64 *ct2,
                        u64 *out) {
    const size_t P = L * n;
    u64 *tmp = (u64 *)malloc(P * sizeof(u64));
    orc_poly_mul(n, L, moduli, ct1, ct2, out);                 /* a0*b0 */
    orc_poly_mul(n, L, moduli, ct1, ct2 + P, out + P);         /* a0*b1 */
    orc_poly_mul(n, L, moduli, ct1 + P, ct2, tmp);             /* a1*b0 */
    orc_poly_add_inplace(n, L, moduli, out + P, tmp);
    orc_poly_mul(n, L, moduli, ct1 + P, ct2 + P, out + 2 * P); /* a1*b1 */
    free(tmp);
}

static int relinearize_common(size_t logn, size_t L, const u64 *moduli_ext, int bgv, u64 t,
                              const u64 *quad, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    const size_t Le = L + 1;
    u64 *ext = (u64 *)malloc(2 * Le * n * sizeof(u64));
    int rc = orc_ext_prod_montgomery(logn, L, moduli_ext, quad + 2 * L * n, key, ext);
    if (rc == 0)
        rc = bgv ? orc_bgv_mod_drop_one_prime(logn, Le, moduli_ext, t, ext, out)
                 : orc_ckks_rescale_by_one_prime(logn, Le, moduli_ext, ext, out);
    if (rc == 0) {
        orc_poly_add_inplace(n, L, moduli_ext, out, quad);
        orc_poly_add_inplace(n, L, moduli_ext, out + L * n, quad + L * n);
    }
    free(ext);
    return rc;
}

/* ckks/arith.cpp:64-73 */
int orc_ckks_relinearize(size_t logn, size_t L, const u64 *moduli_ext, const u64 *quad,
                         const u64 *key, u64 *out) {
    return relinearize_common(logn, L, moduli_ext, 0, 0, quad, key, out);
}

/* bgv/arith.cpp:71-79 (reference behaviour: inner_plain_modulus == 1) */
int orc_bgv_relinearize(size_t logn, size_t L, const u64 *moduli_ext, u64 inner_t,
                        const u64 *quad, const u64 *key, u64 *out) {
    return relinearize_common(logn, L, moduli_ext, 1, inner_t, quad, key, out);
}

/* ckks/arith.cpp:75-93: moved = automorphism(ct); ct' = ext_prod(moved[1], key); drop p; ct'[0] += moved[0] */
static int automorphism_switch(size_t logn, size_t L, const u64 *moduli_ext, int conj, size_t step, const u64 *ct,
                               const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    u64 *moved = (u64 *)malloc(2 * L * n * sizeof(u64));
    u64 *ext = (u64 *)malloc(2 * (L + 1) * n * sizeof(u64));
    for (size_t h = 0; h < 2; h++) {
        if (conj) orc_poly_involution(logn, L, ct + h * L * n, moved + h * L * n);
        else orc_poly_cycle(logn, L, step, ct + h * L * n, moved + h * L * n);
    }
    int rc = orc_ext_prod_montgomery(logn, L, moduli_ext, moved + L * n, key, ext);
    if (rc == 0) rc = orc_ckks_rescale_by_one_prime(logn, L + 1, moduli_ext, ext, out);
    if (rc == 0) orc_poly_add_inplace(n, L, moduli_ext, out, moved);
    free(ext);
    free(moved);
    return rc;
}

int orc_ckks_rotate(size_t logn, size_t L, const u64 *moduli_ext, size_t step, const u64 *ct, const u64 *key, u64 *out) {
    return automorphism_switch(logn, L, moduli_ext, 0, step, ct, key, out);
}

int orc_ckks_conjugate(size_t logn, size_t L, const u64 *moduli_ext, const u64 *ct, const u64 *key, u64 *out) {
    return automorphism_switch(logn, L, moduli_ext, 1, 0, ct, key, out);
}

/* ckks.h:270-274 followed by rescaling.cpp:80-90 */
int orc_ckks_mult_relin_rescale(size_t logn, size_t L, const u64 *moduli_ext, const u64 *ct1,
                                const u64 *ct2, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    u64 *quad = (u64 *)malloc(3 * L * n * sizeof(u64));
    u64 *lin = (u64 *)malloc(2 * L * n * sizeof(u64));
    orc_mult_low_level(n, L, moduli_ext, ct1, ct2, quad);
    int rc = orc_ckks_relinearize(logn, L, moduli_ext, quad, key, lin);
    if (rc == 0) rc = orc_ckks_rescale_by_one_prime(logn, L, moduli_ext, lin, out);
    free(lin);
    free(quad);
    return rc;
}

/* bgv::mult_low_level + bgv::relinearize + bgv::mod_switch_inplace */
int orc_bgv_mult_relin_modswitch(size_t logn, size_t L, const u64 *moduli_ext, u64 t,
                                 const u64 *ct1, const u64 *ct2, const u64 *key, u64 *out) {
    const size_t n = (size_t)1 << logn;
    u64 *quad = (u64 *)malloc(3 * L * n * sizeof(u64));
    u64 *lin = (u64 *)malloc(2 * L * n * sizeof(u64));
    orc_mult_low_level(n, L, moduli_ext, ct1, ct2, quad);
    int rc = orc_bgv_relinearize(logn, L, moduli_ext, 1, quad, key, lin);
    if (rc == 0) rc = orc_bgv_mod_drop_one_prime(logn, L, moduli_ext, t, lin, out);
    free(lin);
    free(quad);
    return rc;
}

/* ===================================================================== */
/* either side of the path: encrypt / decrypt cores, base transforms     */
/* ===================================================================== */

/* sampling.cpp:76-86 lift + NTT, rlwe.cpp:52 (ex - c1*sk), rlwe.cpp:65-69 (NTT(pt), c0 += pt_ntt) */
int orc_rlwe_encrypt_core(size_t logn, size_t L, const u64 *moduli, const int64_t *noise, const u64 *c1,
                          const u64 *pt, const u64 *sk, u64 *ct) {
    const size_t n = (size_t)1 << logn;
    u64 *c0 = ct, *prod = (u64 *)malloc(L * n * sizeof(u64)), *ptn = (u64 *)malloc(L * n * sizeof(u64));
    for (size_t k = 0; k < L; k++) {
        const u64 q = moduli[k];
        for (size_t i = 0; i < n; i++) {
            u64 v = q + (u64)noise[i];
            v -= (v >= q) ? q : 0;
            c0[k * n + i] = v;
        }
    }
    int rc = orc_poly_ntt(logn, L, moduli, c0);
    orc_poly_mul(n, L, moduli, c1, sk, prod);
    orc_poly_sub_inplace(n, L, moduli, c0, prod);
    memcpy(ptn, pt, L * n * sizeof(u64));
    if (rc == 0) rc = orc_poly_ntt(logn, L, moduli, ptn);
    orc_poly_add_inplace(n, L, moduli, c0, ptn);
    memcpy(ct + L * n, c1, L * n * sizeof(u64));
    free(prod);
    free(ptn);
    return rc;
}

/* rlwe.cpp:74-81 */
int orc_rlwe_decrypt_core(size_t logn, size_t L, const u64 *moduli, const u64 *ct, const u64 *sk, u64 *pt) {
    const size_t n = (size_t)1 << logn;
    u64 *prod = (u64 *)malloc(L * n * sizeof(u64));
    orc_poly_mul(n, L, moduli, ct + L * n, sk, prod);
    memcpy(pt, ct, L * n * sizeof(u64));
    orc_poly_add_inplace(n, L, moduli, pt, prod);
    int rc = orc_poly_intt(logn, L, moduli, pt);
    orc_poly_reduce_strict(n, L, moduli, pt);
    free(prod);
    return rc;
}

/* rns_transform.cpp:11-37 (input strictly reduced by :113 first) */
void orc_rns_base_from_single(size_t n, u64 old_modulus, size_t L, const u64 *new_moduli, const u64 *in, u64 *out) {
    const u64 half = old_modulus / 2;
    for (size_t k = 0; k < L; k++) {
        const u64 q = new_moduli[k];
        const u64 multiple = (old_modulus / q + 1) * q;
        for (size_t i = 0; i < n; i++) {
            u64 x = in[i];
            x -= (x >= old_modulus) ? old_modulus : 0;
            out[k * n + i] = (x < half) ? x : multiple - old_modulus + x;
        }
        if (q < old_modulus) orc_batched_barrett_lazy(q, n, out + k * n);
    }
}

int orc_rns_base_to_single_small(size_t n, size_t L, const u64 *old_moduli, u64 new_modulus, const u64 *in, u64 *out);

/* rns_transform.cpp:39-84 */
int orc_rns_base_to_single_small(size_t n, size_t L, const u64 *old_moduli, u64 new_modulus, const u64 *in, u64 *out) {
    const u64 q0 = old_moduli[0], half = q0 / 2;
    int small = 1;
    for (size_t i = 0; i < n && small; i++) {
        u64 x0 = in[i];
        x0 -= (x0 >= q0) ? q0 : 0;
        for (size_t k = 1; k < L; k++) {
            u64 xk = in[k * n + i];
            xk -= (xk >= old_moduli[k]) ? old_moduli[k] : 0;
            if (x0 < half ? (xk != x0) : (old_moduli[k] - xk != q0 - x0)) { small = 0; break; }
        }
    }
    if (!small) return 0;
    const u64 multiple = (q0 / new_modulus + 1) * new_modulus;
    for (size_t i = 0; i < n; i++) {
        u64 x0 = in[i];
        x0 -= (x0 >= q0) ? q0 : 0;
        out[i] = (x0 < half) ? x0 : multiple - q0 + x0;
    }
    orc_batched_barrett(new_modulus, n, out);
    return 1;
}

/* ---- a minimal fixed-width big integer (little-endian u64 words) for the CRT branch ---- */
#define BN_WORDS 17 /* 16 moduli of < 2^64 plus headroom for the sum of L terms */
typedef struct { u64 w[BN_WORDS]; } bn_t;

static void bn_set(bn_t *a, u64 v) { memset(a, 0, sizeof(*a)); a->w[0] = v; }
static void bn_mul_small(bn_t *a, u64 m) {
    u64 carry = 0;
    for (int i = 0; i < BN_WORDS; i++) {
        u128 p = (u128)a->w[i] * m + carry;
        a->w[i] = (u64)p;
        carry = (u64)(p >> 64);
    }
}
static void bn_add(bn_t *a, const bn_t *b) {
    u64 carry = 0;
    for (int i = 0; i < BN_WORDS; i++) {
        u128 s = (u128)a->w[i] + b->w[i] + carry;
        a->w[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
}
static void bn_sub(bn_t *a, const bn_t *b) { /* a >= b */
    u64 borrow = 0;
    for (int i = 0; i < BN_WORDS; i++) {
        u128 d = (u128)a->w[i] - b->w[i] - borrow;
        a->w[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static int bn_cmp(const bn_t *a, const bn_t *b) {
    for (int i = BN_WORDS - 1; i >= 0; i--)
        if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
    return 0;
}
static u64 bn_mod_small(const bn_t *a, u64 m) {
    u128 r = 0;
    for (int i = BN_WORDS - 1; i >= 0; i--) r = ((r << 64) | a->w[i]) % m;
    return (u64)r;
}
static void bn_shr1(bn_t *a) {
    for (int i = 0; i < BN_WORDS; i++) a->w[i] = (a->w[i] >> 1) | (i + 1 < BN_WORDS ? a->w[i + 1] << 63 : 0);
}

/* rns_transform.cpp:106-127 for one new modulus: :113 reduce_strict, then :39-84 when every coefficient is small,
 * else :86-104 (CRT composition, centred around Q/2; note that the second case returns new_modulus itself -- not 0 --
 * when Q - x is a multiple of it) */
void orc_rns_base_to_single(size_t n, size_t L, const u64 *old_moduli, u64 new_modulus, const u64 *in, u64 *out) {
    if (orc_rns_base_to_single_small(n, L, old_moduli, new_modulus, in, out)) return;
    bn_t Q, half, Mi[16];
    u64 ci[16]; /* (Q/q_i)^-1 mod q_i */
    bn_set(&Q, 1);
    for (size_t i = 0; i < L; i++) bn_mul_small(&Q, old_moduli[i]);
    half = Q;
    bn_shr1(&half);
    for (size_t i = 0; i < L; i++) {
        bn_set(&Mi[i], 1);
        for (size_t j = 0; j < L; j++)
            if (j != i) bn_mul_small(&Mi[i], old_moduli[j]);
        ci[i] = orc_inverse_mod_prime(bn_mod_small(&Mi[i], old_moduli[i]), old_moduli[i]);
    }
    for (size_t c = 0; c < n; c++) {
        bn_t big, term;
        bn_set(&big, 0);
        for (size_t i = 0; i < L; i++) {
            u64 x = in[i * n + c];
            x -= (x >= old_moduli[i]) ? old_moduli[i] : 0;
            term = Mi[i];
            bn_mul_small(&term, (u64)((u128)x * ci[i] % old_moduli[i]));
            bn_add(&big, &term);
        }
        while (bn_cmp(&big, &Q) >= 0) bn_sub(&big, &Q);
        if (bn_cmp(&big, &half) < 0) {
            out[c] = bn_mod_small(&big, new_modulus);
        } else {
            bn_t abs = Q;
            bn_sub(&abs, &big);
            out[c] = new_modulus - bn_mod_small(&abs, new_modulus);
        }
    }
}


/* ===================================================================== */
/* digests / generators                                                  */
/* ===================================================================== */

u64 orc_fnv1a64(const void *bytes, size_t nbytes) {
    const unsigned char *p = (const unsigned char *)bytes;
    u64 h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

void orc_splitmix_fill(u64 *state, u64 q, size_t n, u64 *x) {
    u64 s = *state;
    for (size_t i = 0; i < n; i++) {
        u64 z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        x[i] = q ? z % q : z;
    }
    *state = s;
}
