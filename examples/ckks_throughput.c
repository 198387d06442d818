/* ckks_throughput.c -- the throughput path of INTEGRATION.md section 4 in plain C: no Python, no torch, only
 * include/hehub_amd.h.  Uploads a batch of synthetic ciphertext pairs and one relinearisation key, runs
 * ckks::mult + relinearize + rescale_inplace on the whole batch and prints the rate.
 *
 *   gcc -O2 -std=c99 examples/ckks_throughput.c -Iinclude -Lhehub_amd/lib -lhehub_amd \
 *       -Wl,-rpath,$PWD/hehub_amd/lib -o examples/ckks_throughput
 *   examples/ckks_throughput [log2 N = 15] [batch = 64] [steps = 5] [parity level: B (default) | A]
 * Moduli: CKKS parameters {50, 40 x 9} bits + one 50-bit special prime (the BASELINE config 3 chain). */
#define _POSIX_C_SOURCE 199309L
#include "hehub_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static const uint64_t MODULI_EXT[11] = {1125899903827969ull, 1099510054913ull, 1099507695617ull, 1099506515969ull,
                                        1099504549889ull,    1099503894529ull, 1099503370241ull, 1099502714881ull,
                                        1099502518273ull,    1099501731841ull, 1125899904679937ull};

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != HP_OK) {                                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? hp_last_error(ctx) : "");  \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* device array of `rows` limbs of n words, limb r reduced modulo moduli[r % period] */
static int upload_random(hp_ctx *ctx, uint64_t **dptr, size_t rows, size_t n, const uint64_t *moduli, size_t period,
                         uint64_t seed) {
    uint64_t *host = (uint64_t *)malloc(rows * n * sizeof(uint64_t));
    if (!host) return HP_ENOMEM;
    for (size_t r = 0; r < rows; r++)
        for (size_t i = 0; i < n; i++) host[r * n + i] = splitmix(&seed) % moduli[r % period];
    int rc = hp_dev_alloc(ctx, rows * n * sizeof(uint64_t), (void **)dptr);
    if (rc == HP_OK) rc = hp_memcpy_h2d(ctx, *dptr, host, rows * n * sizeof(uint64_t));
    free(host);
    return rc;
}

int main(int argc, char **argv) {
    const size_t logn = argc > 1 ? (size_t)atoi(argv[1]) : 15, batch = argc > 2 ? (size_t)atoi(argv[2]) : 64;
    const int steps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t n = (size_t)1 << logn, L = 10;
    const int level_a = argc > 4 && (argv[4][0] == 'A' || argv[4][0] == 'a');
    hp_ctx *ctx = NULL;
    CHECK(hp_ctx_create(0, &ctx));
    /* level B: every word is hehub's lazy word; level A: its canonical residue (reduce_strict of it), through the FP64 transforms */
    CHECK(hp_ctx_set_parity_level(ctx, level_a ? HP_PARITY_A : HP_PARITY_B));
    printf("%s: N=%zu, L=%zu moduli + special prime, batch %zu, parity level %c\n", hp_version(), n, L, batch, level_a ? 'A' : 'B');

    uint64_t *ct1, *ct2, *key, *out;
    CHECK(upload_random(ctx, &ct1, batch * 2 * L, n, MODULI_EXT, L, 1));           /* u64[batch][2][L][N]   */
    CHECK(upload_random(ctx, &ct2, batch * 2 * L, n, MODULI_EXT, L, 2));
    CHECK(upload_random(ctx, &key, L * 2 * (L + 1), n, MODULI_EXT, L + 1, 3));     /* u64[L][2][L+1][N]     */
    CHECK(hp_dev_alloc(ctx, batch * 2 * (L - 1) * n * sizeof(uint64_t), (void **)&out));

    CHECK(hp_dev_ckks_mult_relin_rescale(ctx, logn, L, MODULI_EXT, batch, ct1, ct2, key, out));   /* warm-up: tables */
    CHECK(hp_sync(ctx));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int s = 0; s < steps; s++) CHECK(hp_dev_ckks_mult_relin_rescale(ctx, logn, L, MODULI_EXT, batch, ct1, ct2, key, out));
    CHECK(hp_sync(ctx));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    uint64_t first[2];
    CHECK(hp_memcpy_d2h(ctx, first, out, sizeof(first)));
    printf("%.1f hom-mult/s (%.3f ms per batch), out[0..1] = %llu %llu\n", (double)batch * steps / sec, 1e3 * sec / steps,
           (unsigned long long)first[0], (unsigned long long)first[1]);
    hp_dev_free(ctx, ct1); hp_dev_free(ctx, ct2); hp_dev_free(ctx, key); hp_dev_free(ctx, out);
    hp_ctx_destroy(ctx);
    return 0;
}
