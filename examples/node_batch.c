/* node_batch.c -- a host-resident batch of ciphertext pairs multiplied on all GPUs of a node through the C ABI alone
 * (include/hehub_amd.h, "node"): batch-sharded mode first, then the same batch through the limb-sharded mode; both must
 * agree word for word with one another and with a single context.  Plain C99, no Python, no torch.
 *
 *   gcc -O2 -std=c99 examples/node_batch.c -Iinclude -Lhehub_amd/lib -lhehub_amd -Wl,-rpath,$PWD/hehub_amd/lib -o examples/node_batch
 *   examples/node_batch [ranks = 2] [gpus = 1] [log2 N = 12] [batch = 6]       (ranks share GPUs when ranks > gpus) */
#include "hehub_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { L = 4 };
static const uint64_t MODULI_EXT[L + 1] = {1099510054913ull, 1099507695617ull, 1099506515969ull, 1099504549889ull,
                                           1125899904679937ull};

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t *random_rows(size_t rows, size_t n, size_t period, uint64_t seed) {
    uint64_t *h = (uint64_t *)malloc(rows * n * sizeof(uint64_t));
    for (size_t r = 0; h && r < rows; r++)
        for (size_t i = 0; i < n; i++) h[r * n + i] = splitmix(&seed) % MODULI_EXT[r % period];
    return h;
}
#define NODE_TRY(call)                                                                   \
    do {                                                                                 \
        int rc__ = (call);                                                               \
        if (rc__ != HP_OK) {                                                             \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, hp_node_last_error(node)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char **argv) {
    const int ranks = argc > 1 ? atoi(argv[1]) : 2, gpus = argc > 2 ? atoi(argv[2]) : 1;
    const size_t logn = argc > 3 ? (size_t)atoi(argv[3]) : 12, batch = argc > 4 ? (size_t)atoi(argv[4]) : 6;
    const size_t n = (size_t)1 << logn, in_words = batch * 2 * L * n, out_words = batch * 2 * (L - 1) * n;
    if (ranks < 1 || ranks > 64 || gpus < 1) return 2;
    int devices[64];
    for (int r = 0; r < ranks; r++) devices[r] = r % gpus;
    hp_node *node = NULL;
    if (hp_node_create(devices, (size_t)ranks, &node) != HP_OK) { fprintf(stderr, "hp_node_create failed (no GPU?)\n"); return 1; }

    uint64_t *ct1 = random_rows(batch * 2 * L, n, L, 1), *ct2 = random_rows(batch * 2 * L, n, L, 2);
    uint64_t *key = random_rows(L * 2 * (L + 1), n, L + 1, 3);
    uint64_t *out_a = (uint64_t *)malloc(out_words * 8), *out_b = (uint64_t *)malloc(out_words * 8), *out_c = (uint64_t *)malloc(out_words * 8);
    uint64_t *d_key[64];
    if (!ct1 || !ct2 || !key || !out_a || !out_b || !out_c) return 1;
    NODE_TRY(hp_node_replicate(node, key, L * 2 * (L + 1) * n, d_key));

    /* 1. batch-sharded: contiguous slices, no exchange */
    NODE_TRY(hp_node_ckks_mult_relin_rescale(node, logn, L, MODULI_EXT, batch, ct1, ct2, d_key, out_a));
    for (int r = 0; r < ranks; r++) {
        size_t lo, hi;
        NODE_TRY(hp_node_slice(node, batch, (size_t)r, &lo, &hi));
        printf("rank %d (gpu %d): ciphertexts [%zu, %zu)\n", r, devices[r], lo, hi);
    }
    /* 2. limb-sharded: every rank works on the whole batch, cut by output modulus, direct peer writes in between */
    hp_node_sharded *plan = NULL;
    NODE_TRY(hp_node_sharded_create(node, logn, L, MODULI_EXT, 0, batch, &plan));
    for (int r = 0; r < ranks; r++) {
        size_t k0, k1;
        NODE_TRY(hp_node_sharded_range(plan, (size_t)r, &k0, &k1));
        printf("rank %d owns extended moduli [%zu, %zu)\n", r, k0, k1);
    }
    NODE_TRY(hp_node_sharded_mult(plan, ct1, ct2, d_key, out_b));
    hp_node_sharded_destroy(plan);
    /* 3. one context, whole batch */
    {
        hp_ctx *ctx = hp_node_ctx(node, 0);
        void *a = NULL, *b = NULL, *o = NULL;
        if (hp_dev_alloc(ctx, in_words * 8, &a) || hp_dev_alloc(ctx, in_words * 8, &b) || hp_dev_alloc(ctx, out_words * 8, &o)) return 1;
        if (hp_memcpy_h2d(ctx, a, ct1, in_words * 8) || hp_memcpy_h2d(ctx, b, ct2, in_words * 8)) return 1;
        if (hp_dev_ckks_mult_relin_rescale(ctx, logn, L, MODULI_EXT, batch, (const uint64_t *)a, (const uint64_t *)b, d_key[0], (uint64_t *)o)) {
            fprintf(stderr, "single context: %s\n", hp_last_error(ctx));
            return 1;
        }
        if (hp_memcpy_d2h(ctx, out_c, o, out_words * 8)) return 1;
        hp_dev_free(ctx, a); hp_dev_free(ctx, b); hp_dev_free(ctx, o);
    }
    const int same = memcmp(out_a, out_b, out_words * 8) == 0 && memcmp(out_a, out_c, out_words * 8) == 0;
    printf("batch-sharded == limb-sharded == single context: %s (fnv %016llx)\n", same ? "yes" : "NO",
           (unsigned long long)hp_wire_fnv1a64(out_a, out_words * 8));
    NODE_TRY(hp_node_free_replicas(node, d_key));
    hp_node_destroy(node);
    free(ct1); free(ct2); free(key); free(out_a); free(out_b); free(out_c);
    return same ? 0 : 1;
}
