/*
 * oracle/mt_harness.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Runs a libaec-ABI implementation (in practice oracle/_ref/libaec_ref.so, the reference compiled
 * from its own sources) on every host core: the input is cut into contiguous RSI-aligned shards,
 * one independent stream per thread, aec_buffer_encode followed by aec_buffer_decode, verified
 * with memcmp.  This is SURVEY.md section 8(d)(ii): "the reference has no threading, so this is
 * the fairest multi-core use".  The two entry points are passed in as function pointers so that
 * this file depends on nothing but the public ABI (include/libaec.h).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/libaec.h"

typedef int (*buffer_fn)(struct aec_stream *);

struct job {
    buffer_fn enc, dec;
    const unsigned char *in;
    size_t n;
    unsigned reps;
    unsigned bps, bs, rsi, flags;
    pthread_barrier_t *start;
    int ok;
};

static void *worker(void *arg)
{
    struct job *j = (struct job *)arg;
    size_t cap = j->n + j->n / 4 + 4096;
    unsigned char *out = (unsigned char *)malloc(cap);
    unsigned char *back = (unsigned char *)malloc(j->n + 4096);
    j->ok = out && back;
    if (j->ok) {                       /* touch the pages before the clock starts */
        memset(out, 0, cap);
        memset(back, 0, j->n + 4096);
    }
    pthread_barrier_wait(j->start);
    for (unsigned r = 0; j->ok && r < j->reps; r++) {
        struct aec_stream s;
        memset(&s, 0, sizeof s);
        s.bits_per_sample = j->bps; s.block_size = j->bs; s.rsi = j->rsi; s.flags = j->flags;
        s.next_in = j->in; s.avail_in = j->n; s.next_out = out; s.avail_out = cap;
        if (j->enc(&s) != AEC_OK) { j->ok = 0; break; }
        size_t clen = s.total_out;
        memset(&s, 0, sizeof s);
        s.bits_per_sample = j->bps; s.block_size = j->bs; s.rsi = j->rsi; s.flags = j->flags;
        s.next_in = out; s.avail_in = clen; s.next_out = back; s.avail_out = j->n;
        if (j->dec(&s) != AEC_OK || memcmp(back, j->in, j->n) != 0) j->ok = 0;
    }
    free(out);
    free(back);
    return NULL;
}

/* Returns 0 and the wall-clock seconds of `reps` encode+decode passes on `threads` shards of
 * `per` bytes each (shard i = in[i*per, (i+1)*per)), or -1 when a thread failed. */
int aec_mt_run(void *enc_fn, void *dec_fn, const unsigned char *in, size_t per, unsigned threads,
               unsigned reps, unsigned bps, unsigned bs, unsigned rsi, unsigned flags, double *seconds)
{
    pthread_t *tid = (pthread_t *)calloc(threads, sizeof *tid);
    struct job *jobs = (struct job *)calloc(threads, sizeof *jobs);
    pthread_barrier_t start;
    if (!tid || !jobs || pthread_barrier_init(&start, NULL, threads + 1)) return -1;
    for (unsigned i = 0; i < threads; i++) {
        struct job j = {(buffer_fn)enc_fn, (buffer_fn)dec_fn, in + (size_t)i * per, per, reps,
                        bps, bs, rsi, flags, &start, 0};
        jobs[i] = j;
        if (pthread_create(&tid[i], NULL, worker, &jobs[i])) return -1;
    }
    struct timespec a, b;
    pthread_barrier_wait(&start);
    clock_gettime(CLOCK_MONOTONIC, &a);
    int ok = 1;
    for (unsigned i = 0; i < threads; i++) {
        pthread_join(tid[i], NULL);
        ok &= jobs[i].ok;
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    *seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    pthread_barrier_destroy(&start);
    free(tid);
    free(jobs);
    return ok ? 0 : -1;
}
