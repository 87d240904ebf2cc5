/*
 * datagen.c -- deterministic synthetic inputs for the benchmark configurations
 * (SURVEY.md section 8(d)).  Host-side utility used by bench.py and the tests to fill
 * buffers before anything is timed; it is not on the codec path.
 *
 * PRNG: SplitMix64, state s0 = 0x5EED0000 + shard, exactly one draw per sample.
 *   lowent16  (configs 2/4)  u16 little endian, x0 = 32768.  At every sample index that is a
 *             multiple of 4096, if (r & 7) == 0 a hold of 256 samples starts (constant run ->
 *             zero blocks, including "rest of segment" runs).  Step g = ctz(r | 2^20)
 *             (geometric, p = 1/2), sign = top bit of r, clamp to [0, 65535].
 *   lowent32s (config 3)     i32 big endian, x0 = 0, g = ctz(r | 2^20) << ((r >> 58) & 7),
 *             clamp to int32, no holds.
 *   chunks8   (config 5)     u8, x0 = 128, g = ctz(r | 2^20), clamp to [0, 255], no holds.
 * The walk is sequential by construction; state is carried in aec_gen_state so a buffer can
 * be produced in pieces.
 */
#include <stddef.h>
#include <stdint.h>

typedef struct aec_gen_state {
    uint64_t s;      /* SplitMix64 state */
    int64_t x;       /* current sample value */
    uint64_t i;      /* sample index */
    uint32_t hold;   /* remaining held samples (lowent16) */
    uint32_t kind;   /* 0 lowent16, 1 lowent32s, 2 chunks8 */
} aec_gen_state;

static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += UINT64_C(0x9E3779B97F4A7C15));
    z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
    return z ^ (z >> 31);
}

void aec_gen_init(aec_gen_state *st, unsigned kind, uint64_t shard)
{
    st->s = UINT64_C(0x5EED0000) + shard;
    st->i = 0;
    st->hold = 0;
    st->kind = kind;
    st->x = kind == 0 ? 32768 : (kind == 1 ? 0 : 128);
}

static inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

/* n = number of SAMPLES; out must hold n * {2,4,1} bytes for kind {0,1,2}. */
void aec_gen_fill(aec_gen_state *st, uint8_t *out, size_t n)
{
    size_t j;
    for (j = 0; j < n; j++, st->i++) {
        uint64_t r = splitmix64(&st->s);
        int64_t g = __builtin_ctzll(r | (UINT64_C(1) << 20));
        int neg = (int)(r >> 63);
        switch (st->kind) {
        case 0:
            if ((st->i & 4095) == 0 && (r & 7) == 0)
                st->hold = 256;
            if (st->hold > 0)
                st->hold--;
            else
                st->x = clamp64(st->x + (neg ? -g : g), 0, 65535);
            out[2 * j] = (uint8_t)st->x;
            out[2 * j + 1] = (uint8_t)(st->x >> 8);
            break;
        case 1:
            g <<= (r >> 58) & 7;
            st->x = clamp64(st->x + (neg ? -g : g), INT32_MIN, INT32_MAX);
            out[4 * j] = (uint8_t)((uint32_t)st->x >> 24);
            out[4 * j + 1] = (uint8_t)((uint32_t)st->x >> 16);
            out[4 * j + 2] = (uint8_t)((uint32_t)st->x >> 8);
            out[4 * j + 3] = (uint8_t)st->x;
            break;
        default:
            st->x = clamp64(st->x + (neg ? -g : g), 0, 255);
            out[j] = (uint8_t)st->x;
            break;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Parallel fill.  The walk is x <- clamp(x + step, lo, hi); maps of that shape are closed under
 * composition (x -> min(max(x + A, L), H)), so each thread first folds its slice into (A, L, H),
 * the slice start values follow from a short serial pass, and a second parallel pass writes
 * the samples.  Slices start at multiples of 4096 samples, where no hold is ever active
 * (holds start at multiples of 4096 and last 256 samples).  SplitMix64 is counter based, so a
 * slice's generator state is s0 + first_index * gamma.  Output is identical to aec_gen_fill.
 * ------------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <stdlib.h>

typedef struct {
    unsigned kind;
    uint64_t shard, first, count;
    uint8_t *out;
    int64_t x0;          /* slice start value (pass 2) */
    int64_t A, L, H;     /* folded map (pass 1) */
    int pass;
} gen_job;

static void gen_limits(unsigned kind, int64_t *lo, int64_t *hi)
{
    if (kind == 0) { *lo = 0; *hi = 65535; }
    else if (kind == 1) { *lo = INT32_MIN; *hi = INT32_MAX; }
    else { *lo = 0; *hi = 255; }
}

static void *gen_worker(void *arg)
{
    gen_job *j = (gen_job *)arg;
    if (j->pass == 2) {
        aec_gen_state st;
        aec_gen_init(&st, j->kind, j->shard);
        st.s += j->first * UINT64_C(0x9E3779B97F4A7C15);
        st.i = j->first;
        st.x = j->x0;
        aec_gen_fill(&st, j->out, j->count);
        return NULL;
    }
    {
        int64_t lo, hi, A = 0, L = INT64_MIN / 4, H = INT64_MAX / 4;
        uint64_t s = UINT64_C(0x5EED0000) + j->shard + j->first * UINT64_C(0x9E3779B97F4A7C15);
        uint32_t hold = 0;
        uint64_t i;
        gen_limits(j->kind, &lo, &hi);
        for (i = j->first; i < j->first + j->count; i++) {
            uint64_t r = splitmix64(&s);
            int64_t g = __builtin_ctzll(r | (UINT64_C(1) << 20));
            if (j->kind == 0) {
                if ((i & 4095) == 0 && (r & 7) == 0) hold = 256;
                if (hold > 0) { hold--; continue; }
            } else if (j->kind == 1) {
                g <<= (r >> 58) & 7;
            }
            if (r >> 63) g = -g;
            A += g;
            L = clamp64(L + g, lo, hi);
            H = clamp64(H + g, lo, hi);
        }
        j->A = A; j->L = L; j->H = H;
    }
    return NULL;
}

/* n = number of samples; bytes per sample {2,4,1} for kind {0,1,2}. */
void aec_gen_fill_parallel(unsigned kind, uint64_t shard, uint8_t *out, size_t n, unsigned nthreads)
{
    const size_t bps = kind == 0 ? 2 : (kind == 1 ? 4 : 1);
    size_t per, t, nj;
    gen_job *jobs;
    pthread_t *th;
    int64_t x;
    aec_gen_state st0;

    if (nthreads < 1) nthreads = 1;
    per = (n / nthreads + 4095) & ~(size_t)4095;
    if (per == 0) per = 4096;
    nj = (n + per - 1) / per;
    if (nj <= 1) {
        aec_gen_init(&st0, kind, shard);
        aec_gen_fill(&st0, out, n);
        return;
    }
    jobs = (gen_job *)calloc(nj, sizeof *jobs);
    th = (pthread_t *)calloc(nj, sizeof *th);
    for (t = 0; t < nj; t++) {
        jobs[t].kind = kind; jobs[t].shard = shard;
        jobs[t].first = t * per;
        jobs[t].count = (t + 1) * per <= n ? per : n - t * per;
        jobs[t].out = out + t * per * bps;
        jobs[t].pass = 1;
        pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
    }
    for (t = 0; t < nj; t++) pthread_join(th[t], NULL);
    aec_gen_init(&st0, kind, shard);
    x = st0.x;
    for (t = 0; t < nj; t++) {
        int64_t y = x + jobs[t].A;
        jobs[t].x0 = x;
        if (y < jobs[t].L) y = jobs[t].L;
        if (y > jobs[t].H) y = jobs[t].H;
        x = y;
        jobs[t].pass = 2;
    }
    for (t = 0; t < nj; t++) pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
    for (t = 0; t < nj; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}
