/* multi_ctx.c -- one engine context per host thread (include/hehub_amd.h: "one hp_ctx per GPU ... different ctxs are
 * independent").  Thread i creates its own context on GPU (i mod #gpus), uploads the same synthetic ciphertext pairs
 * and key, runs ckks::mult + relinearize + rescale_inplace on its batch and returns a digest of the result; all
 * digests must agree.  On an 8-GPU node this is the batch-sharded mode without Python (no collective is needed:
 * SURVEY.md 8e); on a 1-GPU box it checks that several contexts can share a device concurrently.
 *
 *   gcc -O2 -std=c99 -pthread examples/multi_ctx.c -Iinclude -Lhehub_amd/lib -lhehub_amd \
 *       -Wl,-rpath,$PWD/hehub_amd/lib -o examples/multi_ctx
 *   examples/multi_ctx [threads = 2] [gpus = 1] [log2 N = 13] [batch = 4] [steps = 3] */
#define _POSIX_C_SOURCE 199309L
#include "hehub_amd.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

enum { L = 4 };
static const uint64_t MODULI_EXT[L + 1] = {1099510054913ull, 1099507695617ull, 1099506515969ull, 1099504549889ull,
                                           1125899904679937ull};

typedef struct {
    int id, gpu, steps, rc;
    size_t logn, batch;
    uint64_t digest;
    double seconds;
    char err[256];
} job_t;

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int upload_random(hp_ctx *ctx, uint64_t **dptr, size_t rows, size_t n, size_t period, uint64_t seed) {
    uint64_t *host = (uint64_t *)malloc(rows * n * sizeof(uint64_t));
    if (!host) return HP_ENOMEM;
    for (size_t r = 0; r < rows; r++)
        for (size_t i = 0; i < n; i++) host[r * n + i] = splitmix(&seed) % MODULI_EXT[r % period];
    int rc = hp_dev_alloc(ctx, rows * n * sizeof(uint64_t), (void **)dptr);
    if (rc == HP_OK) rc = hp_memcpy_h2d(ctx, *dptr, host, rows * n * sizeof(uint64_t));
    free(host);
    return rc;
}

#define TRY(call)                                                                        \
    do {                                                                                 \
        j->rc = (call);                                                                  \
        if (j->rc != HP_OK) {                                                            \
            snprintf(j->err, sizeof j->err, "%s: %s", #call, ctx ? hp_last_error(ctx) : ""); \
            goto done;                                                                   \
        }                                                                                \
    } while (0)

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    const size_t n = (size_t)1 << j->logn, words_out = j->batch * 2 * (L - 1) * n;
    hp_ctx *ctx = NULL;
    uint64_t *ct1 = NULL, *ct2 = NULL, *key = NULL, *out = NULL, *host = NULL;
    TRY(hp_ctx_create(j->gpu, &ctx));
    TRY(upload_random(ctx, &ct1, j->batch * 2 * L, n, L, 1));
    TRY(upload_random(ctx, &ct2, j->batch * 2 * L, n, L, 2));
    TRY(upload_random(ctx, &key, L * 2 * (L + 1), n, L + 1, 3));
    TRY(hp_dev_alloc(ctx, words_out * sizeof(uint64_t), (void **)&out));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int s = 0; s < j->steps; s++) TRY(hp_dev_ckks_mult_relin_rescale(ctx, j->logn, L, MODULI_EXT, j->batch, ct1, ct2, key, out));
    TRY(hp_sync(ctx));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    j->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    host = (uint64_t *)malloc(words_out * sizeof(uint64_t));
    if (!host) { j->rc = HP_ENOMEM; goto done; }
    TRY(hp_memcpy_d2h(ctx, host, out, words_out * sizeof(uint64_t)));
    j->digest = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < words_out; i++) j->digest = (j->digest ^ host[i]) * 0x100000001b3ull;
done:
    free(host);
    if (ctx) {
        hp_dev_free(ctx, ct1); hp_dev_free(ctx, ct2); hp_dev_free(ctx, key); hp_dev_free(ctx, out);
        hp_ctx_destroy(ctx);
    }
    return NULL;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 2, gpus = argc > 2 ? atoi(argv[2]) : 1;
    const size_t logn = argc > 3 ? (size_t)atoi(argv[3]) : 13, batch = argc > 4 ? (size_t)atoi(argv[4]) : 4;
    const int steps = argc > 5 ? atoi(argv[5]) : 3;
    if (threads < 1 || threads > 64 || gpus < 1) return 2;
    pthread_t tid[64];
    job_t jobs[64];
    for (int i = 0; i < threads; i++) {
        jobs[i] = (job_t){.id = i, .gpu = i % gpus, .steps = steps, .rc = 0, .logn = logn, .batch = batch};
        if (pthread_create(&tid[i], NULL, worker, &jobs[i]) != 0) return 3;
    }
    int bad = 0;
    double ops = 0, slowest = 0;
    for (int i = 0; i < threads; i++) {
        pthread_join(tid[i], NULL);
        if (jobs[i].rc != HP_OK) { fprintf(stderr, "thread %d failed (%d): %s\n", i, jobs[i].rc, jobs[i].err); bad++; continue; }
        if (jobs[i].digest != jobs[0].digest) { fprintf(stderr, "thread %d: digest differs\n", i); bad++; }
        ops += (double)batch * steps;
        if (jobs[i].seconds > slowest) slowest = jobs[i].seconds;
    }
    if (!bad) printf("%d contexts on %d GPU(s): digest %016llx everywhere, %.1f hom-mult/s in aggregate\n", threads, gpus,
                     (unsigned long long)jobs[0].digest, ops / slowest);
    return bad ? 1 : 0;
}
