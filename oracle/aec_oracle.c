/*
 * aec_oracle.c -- plain-C restatement of the libaec 0.3.4 encoder and decoder
 * (CCSDS 121.0-B-2), whole-buffer form.  TEST INFRASTRUCTURE ONLY -- see aec_oracle.h.
 *
 * Every function names the reference lines it restates (paths relative to
 * /root/reference).  The code is written as straight loops over RSIs and blocks with
 * one MSB-first bit writer / bit reader; it deliberately keeps the reference's
 * *sequential* formulation of the k search (hill climb carrying state->k from block to
 * block) so that it is an independent check of the parallel reformulation used by the
 * HIP kernels (plateau clamp + scan).
 */
#include "aec_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* parameters                                                                */
/* ------------------------------------------------------------------------- */

/* encode.c:777-872 (encoder) and decode.c:699-766 (decoder; it validates less). */
int aeco_derive(const aeco_params *p, int for_encode, aeco_derived *d)
{
    unsigned bps = p->bits_per_sample;

    if (bps == 0 || bps > 32)
        return AECO_CONF_ERROR;                       /* encode.c:777, decode.c:699 */

    if (for_encode) {
        if (p->flags & AECO_NOT_ENFORCE) {
            if (p->block_size & 1)                    /* encode.c:780-783 */
                return AECO_CONF_ERROR;
        } else if (p->block_size != 8 && p->block_size != 16 &&
                   p->block_size != 32 && p->block_size != 64) {
            return AECO_CONF_ERROR;                   /* encode.c:785-790 */
        }
        if (p->rsi > 4096)
            return AECO_CONF_ERROR;                   /* encode.c:793 */
    }
    if (p->block_size == 0 || p->rsi == 0)
        return AECO_CONF_ERROR;   /* the reference does not test these; both loop forever */

    if (bps > 16) {                                   /* encode.c:804-828 */
        d->id_len = 5;
        d->bytes_per_sample = (bps <= 24 && (p->flags & AECO_3BYTE)) ? 3 : 4;
    } else if (bps > 8) {                             /* encode.c:829-840 */
        d->id_len = 4;
        d->bytes_per_sample = 2;
    } else {                                          /* encode.c:841-859 */
        if (p->flags & AECO_RESTRICTED) {
            if (bps > 4)
                return AECO_CONF_ERROR;
            d->id_len = (bps <= 2) ? 1 : 2;
        } else {
            d->id_len = 3;
        }
        d->bytes_per_sample = 1;
    }

    if (p->flags & AECO_SIGNED) {                     /* encode.c:862-870 */
        d->xmax = UINT32_MAX >> (32 - bps + 1);
        d->xmin = ~d->xmax;
    } else {
        d->xmin = 0;
        d->xmax = UINT32_MAX >> (32 - bps);
    }
    d->kmax = (1 << d->id_len) - 3;                   /* encode.c:872 */
    return AECO_OK;
}

static uint32_t low_mask(int n)
{
    return n >= 32 ? UINT32_MAX : ((UINT32_C(1) << n) - 1);
}

/* ------------------------------------------------------------------------- */
/* sample <-> byte conversions                                               */
/* ------------------------------------------------------------------------- */

/* encode_accessors.c:61-143 (aec_get_8 .. aec_get_msb_32) */
static uint32_t sample_load(const uint8_t *p, int bytes, int msb)
{
    uint32_t v = 0;
    int i;
    if (msb)
        for (i = 0; i < bytes; i++)
            v = (v << 8) | p[i];
    else
        for (i = bytes - 1; i >= 0; i--)
            v = (v << 8) | p[i];
    return v;
}

/* decode.c:144-189 (put_msb_32 .. put_8) */
static void sample_store(uint8_t *p, uint32_t v, int bytes, int msb)
{
    int i;
    if (msb)
        for (i = 0; i < bytes; i++)
            p[i] = (uint8_t)(v >> (8 * (bytes - 1 - i)));
    else
        for (i = 0; i < bytes; i++)
            p[i] = (uint8_t)(v >> (8 * i));
}

/* ------------------------------------------------------------------------- */
/* bit writer (MSB first; encode.c:61-104 emit / emitfs)                     */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint8_t *out;
    size_t cap;
    size_t pos;       /* bytes produced (may exceed cap: overflow is detected at the end) */
    uint64_t acc;     /* pending bits, right aligned */
    int nacc;         /* number of pending bits, always < 8 between calls */
    uint64_t total;   /* stream bits so far */
} bitw;

static void bw_put(bitw *w, uint32_t v, int n)
{
    if (n == 0)
        return;
    w->acc = (w->acc << n) | (uint64_t)(v & low_mask(n));
    w->nacc += n;
    w->total += (uint64_t)n;
    while (w->nacc >= 8) {
        w->nacc -= 8;
        if (w->pos < w->cap)
            w->out[w->pos] = (uint8_t)(w->acc >> w->nacc);
        w->pos++;
    }
    w->acc &= 0xff;
}

/* fundamental sequence: `zeros` 0-bits then a 1 (encode.c:85-104) */
static void bw_unary(bitw *w, uint64_t zeros)
{
    while (zeros >= 32) {
        bw_put(w, 0, 32);
        zeros -= 32;
    }
    bw_put(w, 1, (int)zeros + 1);
}

/* ------------------------------------------------------------------------- */
/* encoder                                                                   */
/* ------------------------------------------------------------------------- */

/* encode.c:235-271 preprocess_unsigned: unit-delay predictor + sign map */
static void pp_unsigned(const uint32_t *x, uint32_t *d, size_t n, uint32_t xmax)
{
    size_t i;
    d[0] = 0;
    for (i = 1; i < n; i++) {
        uint32_t prev = x[i - 1], cur = x[i];
        if (cur >= prev) {
            uint32_t delta = cur - prev;
            d[i] = (delta <= prev) ? 2 * delta : cur;
        } else {
            uint32_t delta = prev - cur;
            d[i] = (delta <= xmax - prev) ? 2 * delta - 1 : xmax - cur;
        }
    }
}

/* encode.c:273-311 preprocess_signed; samples are sign extended from bit bps-1 first */
static void pp_signed(const uint32_t *raw, uint32_t *d, size_t n, unsigned bps,
                      uint32_t uxmin, uint32_t uxmax)
{
    uint32_t m = UINT32_C(1) << (bps - 1);
    int32_t xmin = (int32_t)uxmin, xmax = (int32_t)uxmax;
    int32_t prev = (int32_t)((raw[0] ^ m) - m);
    size_t i;
    d[0] = 0;
    for (i = 1; i < n; i++) {
        int32_t cur = (int32_t)((raw[i] ^ m) - m);
        if (cur < prev) {
            uint32_t delta = (uint32_t)prev - (uint32_t)cur;
            d[i] = (delta <= (uint32_t)xmax - (uint32_t)prev)
                 ? 2 * delta - 1 : (uint32_t)xmax - (uint32_t)cur;
        } else {
            uint32_t delta = (uint32_t)cur - (uint32_t)prev;
            d[i] = (delta <= (uint32_t)prev - (uint32_t)xmin)
                 ? 2 * delta : (uint32_t)cur - (uint32_t)xmin;
        }
        prev = cur;
    }
}

/* encode.c:313-327 block_fs: sum over the WHOLE block (sample 0 of a reference block
 * is the preprocessor's d[0] = 0, so it adds nothing). */
static uint64_t fs_sum(const uint32_t *blk, unsigned bs, int k)
{
    uint64_t s = 0;
    unsigned i;
    for (i = 0; i < bs; i++)
        s += (uint64_t)(blk[i] >> k);
    return s;
}

/*
 * encode.c:329-410 assess_splitting_option.  Hill climb that starts at the k chosen for
 * the previous (non-zero) block, accepts strict improvements only, first tries larger k
 * and turns around at most once.  *kstate is state->k and is rewritten (encode.c:407)
 * whether or not the split option ends up being selected.
 */
static uint32_t split_cost(const uint32_t *blk, unsigned bs, int ref, int kmax, int *kstate)
{
    const uint64_t coded = bs - (unsigned)ref;
    const int kstart = *kstate;
    uint64_t best = UINT64_MAX;
    int kbest = kstart;
    int k = kstart;
    int rising = 1;
    int turn_allowed = (kstart != 0);       /* no_turn = (k == 0), encode.c:370 */

    for (;;) {
        uint64_t fs = fs_sum(blk, bs, k);
        uint64_t len = fs + coded * (uint64_t)(k + 1);
        int go_down = 0;

        if (len < best) {
            if (best != UINT64_MAX)
                turn_allowed = 0;           /* encode.c:378-379 */
            best = len;
            kbest = k;
            if (rising) {
                if (fs < coded || k >= kmax)
                    go_down = 1;            /* encode.c:385-390 */
                else
                    k++;
            } else {
                if (fs >= coded || k == 0)
                    break;                  /* encode.c:395-396 */
                k--;
            }
        } else {
            go_down = 1;                    /* encode.c:399-404 */
        }
        if (go_down) {
            if (!turn_allowed)
                break;
            k = kstart - 1;
            rising = 0;
            turn_allowed = 0;
        }
    }
    *kstate = kbest;
    return (uint32_t)best;                  /* encode.c:409 truncation */
}

/* encode.c:412-434 assess_se_option (all arithmetic in uint64_t, early out) */
static uint32_t se_cost(const uint32_t *blk, unsigned bs, uint32_t limit)
{
    uint64_t len = 1;
    unsigned i;
    for (i = 0; i < bs; i += 2) {
        uint64_t s = (uint64_t)blk[i] + (uint64_t)blk[i + 1];
        len += s * (s + 1) / 2 + blk[i + 1] + 1;
        if (len > limit)
            return UINT32_MAX;
    }
    return (uint32_t)len;
}

typedef struct {
    const aeco_params *p;
    aeco_derived dv;
    bitw w;
    int k;                  /* state->k: survives blocks AND RSIs (never reset) */
    aeco_block_trace *trace;
    size_t blk_index;       /* global block counter for the trace */
} enc_ctx;

static void trace_set(enc_ctx *c, size_t idx, int option, uint64_t bits_before)
{
    if (c->trace) {
        c->trace[idx].option = (uint8_t)option;
        c->trace[idx].k = (uint8_t)c->k;
        c->trace[idx].reserved = 0;
        c->trace[idx].bits = (uint32_t)(c->w.total - bits_before);
    }
}

/* encode.c:565-583 m_encode_zero.  `run` = number of zero blocks, ros = run closed by the
 * end of a segment/RSI with more than 4 blocks (encode.c:649-651). */
static void put_zero_run(enc_ctx *c, int run, int ros, int ref, uint32_t ref_sample)
{
    bw_put(&c->w, 0, c->dv.id_len + 1);
    if (ref)
        bw_put(&c->w, ref_sample, (int)c->p->bits_per_sample);
    if (ros)
        bw_unary(&c->w, 4);
    else if (run >= 5)
        bw_unary(&c->w, (uint64_t)run);
    else
        bw_unary(&c->w, (uint64_t)run - 1);
}

/* encode.c:585-612 m_select_code_option + 520-563 m_encode_{splitting,uncomp,se} */
static int put_block(enc_ctx *c, const uint32_t *blk, int ref, uint32_t ref_sample)
{
    const unsigned bs = c->p->block_size, bps = c->p->bits_per_sample;
    const uint32_t uncomp_len = (bs - (unsigned)ref) * bps;   /* encode.c:270,746 */
    uint32_t split_len, se_len;
    unsigned i;

    if (c->dv.id_len > 1)
        split_len = split_cost(blk, bs, ref, c->dv.kmax, &c->k);
    else
        split_len = UINT32_MAX;                                /* encode.c:595-598 */
    se_len = se_cost(blk, bs, uncomp_len);

    if (split_len < uncomp_len && split_len < se_len) {
        int k = c->k;
        bw_put(&c->w, (uint32_t)k + 1, c->dv.id_len);
        if (ref)
            bw_put(&c->w, ref_sample, (int)bps);
        for (i = (unsigned)ref; i < bs; i++)                   /* encode.c:118-142 */
            bw_unary(&c->w, blk[i] >> k);
        if (k)
            for (i = (unsigned)ref; i < bs; i++)               /* encode.c:144-233 */
                bw_put(&c->w, blk[i], k);
        return AECO_OPT_SPLIT;
    }
    if (split_len >= uncomp_len && uncomp_len <= se_len) {
        bw_put(&c->w, (UINT32_C(1) << c->dv.id_len) - 1, c->dv.id_len);
        bw_put(&c->w, ref ? ref_sample : blk[0], (int)bps);    /* encode.c:541-542 */
        for (i = 1; i < bs; i++)
            bw_put(&c->w, blk[i], (int)bps);
        return AECO_OPT_UNCOMP;
    }
    bw_put(&c->w, 1, c->dv.id_len + 1);                        /* encode.c:553 */
    if (ref)
        bw_put(&c->w, ref_sample, (int)bps);
    for (i = 0; i < bs; i += 2) {
        uint32_t s = blk[i] + blk[i + 1];                      /* encode.c:558 (uint32) */
        bw_unary(&c->w, (uint64_t)(s * (s + 1) / 2 + blk[i + 1]));
    }
    return AECO_OPT_SE;
}

/* One RSI holding `nblk` coded blocks: encode.c:614-659 (zero-block aggregation) around
 * encode.c:709-754 (block dispenser). */
static void put_rsi(enc_ctx *c, const uint32_t *d, unsigned nblk, int pp, uint32_t ref_sample)
{
    const unsigned bs = c->p->block_size;
    int run = 0, run_ref = 0;
    size_t run_first = 0;
    uint64_t mark;
    unsigned b, i;

    for (b = 0; b < nblk; b++) {
        const uint32_t *blk = d + (size_t)b * bs;
        int ref = (pp && b == 0);
        int closes = (b + 1 == nblk) || ((b + 1) % 64 == 0);   /* encode.c:649 */
        int zero = 1;

        for (i = 0; i < bs; i++)
            if (blk[i]) { zero = 0; break; }

        if (zero) {
            if (run++ == 0) {
                run_ref = ref;                                  /* encode.c:645-648 */
                run_first = c->blk_index;
            } else {
                trace_set(c, c->blk_index, AECO_OPT_ZERO_CONT, c->w.total);
            }
            if (closes) {
                mark = c->w.total;
                put_zero_run(c, run, run > 4, run_ref, ref_sample);
                trace_set(c, run_first, AECO_OPT_ZERO, mark);
                run = 0;
            }
        } else {
            if (run) {                                          /* encode.c:632-640 */
                mark = c->w.total;
                put_zero_run(c, run, 0, run_ref, ref_sample);
                trace_set(c, run_first, AECO_OPT_ZERO, mark);
                run = 0;
            }
            mark = c->w.total;
            trace_set(c, c->blk_index, put_block(c, blk, ref, ref_sample), mark);
        }
        c->blk_index++;
    }
}

int aeco_encode(const aeco_params *p, const uint8_t *in, size_t in_len,
                uint8_t *out, size_t out_cap, size_t *out_len,
                aeco_block_trace *trace, uint64_t *rsi_bit_off, uint64_t *total_bits)
{
    enc_ctx c;
    size_t nsamp, rsi_samples, s0, r = 0;
    uint32_t *raw, *d;
    int pp, msb, rc;

    memset(&c, 0, sizeof c);
    rc = aeco_derive(p, 1, &c.dv);
    if (rc != AECO_OK)
        return rc;
    c.p = p;
    c.trace = trace;
    c.w.out = out;
    c.w.cap = out_cap;

    pp = (p->flags & AECO_PREPROCESS) != 0;
    msb = (p->flags & AECO_MSB) != 0;
    nsamp = in_len / (size_t)c.dv.bytes_per_sample;
    rsi_samples = (size_t)p->rsi * p->block_size;

    raw = (uint32_t *)malloc(rsi_samples * sizeof *raw);
    d = (uint32_t *)malloc(rsi_samples * sizeof *d);
    if (!raw || !d) {
        free(raw); free(d);
        return AECO_MEM_ERROR;
    }

    for (s0 = 0; s0 < nsamp; s0 += rsi_samples, r++) {
        size_t n = nsamp - s0 < rsi_samples ? nsamp - s0 : rsi_samples;
        unsigned nblk = (unsigned)((n + p->block_size - 1) / p->block_size);
        uint32_t ref_sample;
        size_t i;

        for (i = 0; i < n; i++)
            raw[i] = sample_load(in + (s0 + i) * (size_t)c.dv.bytes_per_sample,
                                 c.dv.bytes_per_sample, msb);
        for (; i < rsi_samples; i++)       /* encode.c:676-684: repeat last sample */
            raw[i] = raw[n - 1];

        if (rsi_bit_off)
            rsi_bit_off[r] = c.w.total;

        ref_sample = raw[0] & low_mask((int)p->bits_per_sample);   /* encode.c:253,290 */
        if (!pp)
            put_rsi(&c, raw, nblk, 0, 0);  /* data_raw == data_pp, encode.c:891 */
        else {
            if (p->flags & AECO_SIGNED)
                pp_signed(raw, d, rsi_samples, p->bits_per_sample, c.dv.xmin, c.dv.xmax);
            else
                pp_unsigned(raw, d, rsi_samples, c.dv.xmax);
            put_rsi(&c, d, nblk, 1, ref_sample);
        }
    }
    free(raw);
    free(d);

    if (total_bits)
        *total_bits = c.w.total;

    /* encode.c:686-695: the byte under the cursor is always written out, zero padded;
     * an empty stream therefore still yields one 0x00 byte. */
    if (c.w.total == 0 || c.w.nacc > 0) {
        if (c.w.pos < c.w.cap)
            c.w.out[c.w.pos] = (uint8_t)(c.w.acc << (8 - c.w.nacc));
        c.w.pos++;
    }
    if (out_len)
        *out_len = c.w.pos < c.w.cap ? c.w.pos : c.w.cap;
    return c.w.pos <= c.w.cap ? AECO_OK : AECO_STREAM_ERROR;   /* encode.c:944-945 */
}

/* ------------------------------------------------------------------------- */
/* decoder                                                                   */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *in;
    uint64_t nbits;   /* bits in the buffer */
    uint64_t pos;     /* next unread bit */
} bitr;

static int br_has(const bitr *r, int n)
{
    return r->pos + (uint64_t)n <= r->nbits;
}

/* decode.c:222-286 direct_get / 355-364 bits_get+bits_drop */
static uint32_t br_get(bitr *r, int n)
{
    uint64_t v = 0;
    int i;
    for (i = 0; i < n; i++) {
        uint64_t b = r->pos + (uint64_t)i;
        v = (v << 1) | ((r->in[b >> 3] >> (7 - (b & 7))) & 1u);
    }
    r->pos += (uint64_t)n;
    return (uint32_t)v;
}

/* decode.c:288-340 direct_get_fs / 366-389 fs_ask+fs_drop.  Returns 0 (cursor unchanged)
 * when the buffer ends before the terminating 1 bit. */
static int br_unary(bitr *r, uint32_t *zeros)
{
    uint64_t q = r->pos;
    while (q < r->nbits) {
        if ((r->in[q >> 3] >> (7 - (q & 7))) & 1u) {
            *zeros = (uint32_t)(q - r->pos);
            r->pos = q + 1;
            return 1;
        }
        /* skip whole zero bytes quickly */
        if ((q & 7) == 0 && r->in[q >> 3] == 0)
            q += 8;
        else
            q++;
    }
    return 0;
}

typedef struct {
    const aeco_params *p;
    aeco_derived dv;
    bitr r;
    uint32_t *buf;          /* rsi_buffer: decoded (still preprocessed) samples of one RSI */
    size_t filled;          /* samples in buf */
    size_t flushed;         /* samples of buf already post-processed and written */
    int32_t last;           /* last_out */
    uint8_t *out;
    size_t budget;          /* output samples still allowed (avail_out / bytes_per_sample) */
    size_t written;         /* samples written */
} dec_ctx;

/* decode.c:67-141 FLUSH(KIND): inverse predictor fused with the byte-order store */
static void dec_flush(dec_ctx *c)
{
    const int bytes = c->dv.bytes_per_sample;
    const int msb = (c->p->flags & AECO_MSB) != 0;
    size_t i = c->flushed;

    if (!(c->p->flags & AECO_PREPROCESS)) {
        for (; i < c->filled; i++, c->written++)
            sample_store(c->out + c->written * (size_t)bytes, c->buf[i], bytes, msb);
        c->flushed = c->filled;
        return;
    }
    if (i == 0 && c->filled > 0) {                     /* decode.c:76-87 */
        c->last = (int32_t)c->buf[0];
        if (c->p->flags & AECO_SIGNED) {
            uint32_t m = UINT32_C(1) << (c->p->bits_per_sample - 1);
            c->last = (int32_t)(((uint32_t)c->last ^ m) - m);
        }
        sample_store(c->out + c->written * (size_t)bytes, (uint32_t)c->last, bytes, msb);
        c->written++;
        i = 1;
    }
    if (c->dv.xmin == 0) {                             /* decode.c:91-110 */
        uint32_t xmax = c->dv.xmax, med = xmax / 2 + 1;
        uint32_t x = (uint32_t)c->last;
        for (; i < c->filled; i++, c->written++) {
            uint32_t dd = c->buf[i];
            uint32_t half = (dd >> 1) + (dd & 1);
            uint32_t mask = (x & med) ? xmax : 0;
            if (half <= (mask ^ x))
                x += (dd & 1) ? ~(dd >> 1) : (dd >> 1);
            else
                x = mask ^ dd;
            sample_store(c->out + c->written * (size_t)bytes, x, bytes, msb);
        }
        c->last = (int32_t)x;
    } else {                                           /* decode.c:111-135 */
        int32_t xmax = (int32_t)c->dv.xmax;
        int32_t x = c->last;
        for (; i < c->filled; i++, c->written++) {
            uint32_t dd = c->buf[i];
            uint32_t half = (dd >> 1) + (dd & 1);
            uint32_t step = (dd & 1) ? ~(dd >> 1) : (dd >> 1);
            if (x < 0) {
                if (half <= (uint32_t)xmax + (uint32_t)x + 1)
                    x = (int32_t)((uint32_t)x + step);
                else
                    x = (int32_t)(dd - (uint32_t)xmax - 1);
            } else {
                if (half <= (uint32_t)xmax - (uint32_t)x)
                    x = (int32_t)((uint32_t)x + step);
                else
                    x = (int32_t)((uint32_t)xmax - dd);
            }
            sample_store(c->out + c->written * (size_t)bytes, (uint32_t)x, bytes, msb);
        }
        c->last = x;
    }
    c->flushed = c->filled;
}

/* decode.c:213-220 put_sample + 199-211 check_rsi_end; returns 0 when the output is full */
static int dec_emit(dec_ctx *c, uint32_t v)
{
    if (c->budget == 0)
        return 0;
    c->buf[c->filled++] = v;
    c->budget--;
    if (c->filled == (size_t)c->p->rsi * c->p->block_size) {
        dec_flush(c);
        c->filled = 0;
        c->flushed = 0;
    }
    return 1;
}

/* decode.c:391-400 copysample */
static int dec_copy_sample(dec_ctx *c)
{
    int bps = (int)c->p->bits_per_sample;
    if (!br_has(&c->r, bps) || c->budget == 0)
        return 0;
    return dec_emit(c, br_get(&c->r, bps));
}

/* decode.c:679-692 create_se_table, as closed form: for code value m find s = a+b with
 * s(s+1)/2 <= m < (s+1)(s+2)/2; valid table range is m <= 90 (13 rows). */
static int se_split(uint32_t m, uint32_t *sum, uint32_t *base)
{
    uint32_t s = 0, tri = 0;
    if (m > 90)
        return 0;
    while (tri + s + 1 <= m) {
        tri += s + 1;
        s++;
    }
    *sum = s;
    *base = tri;
    return 1;
}

/* Returns 1 when a whole CDS was consumed, 0 when decoding has to stop (input or output
 * exhausted; everything decodable so far has been emitted), <0 on error. */
static int dec_cds(dec_ctx *c)
{
    const unsigned bs = c->p->block_size;
    const int bps = (int)c->p->bits_per_sample;
    const int pp = (c->p->flags & AECO_PREPROCESS) != 0;
    const int ref = pp && c->filled == 0;                    /* decode.c:406-413 */
    uint32_t id, i;

    if (c->filled == 0 && (c->p->flags & AECO_PAD_RSI))      /* decode.c:407-408 */
        c->r.pos = (c->r.pos + 7) & ~UINT64_C(7);

    if (!br_has(&c->r, c->dv.id_len))
        return 0;
    id = br_get(&c->r, c->dv.id_len);

    if (id == 0) {                                           /* decode.c:634-644, 618-632 */
        uint32_t sel, fs;
        if (!br_has(&c->r, 1))
            return 0;
        sel = br_get(&c->r, 1);
        if (ref && !dec_copy_sample(c))
            return 0;
        if (sel == 1) {                                      /* decode.c:560-616 SE */
            i = (uint32_t)ref;
            while (i < bs) {
                uint32_t s, base, second;
                if (!br_unary(&c->r, &fs))
                    return 0;
                if (!se_split(fs, &s, &base))
                    return AECO_DATA_ERROR;   /* reference reads past se_table here */
                second = fs - base;
                if ((i & 1) == 0) {
                    if (!dec_emit(c, s - second))
                        return 0;
                    i++;
                }
                if (!dec_emit(c, second))
                    return 0;
                i++;
            }
            return 1;
        } else {                                             /* decode.c:518-558 zero run */
            uint32_t nzb, b, count;
            if (!br_unary(&c->r, &fs))
                return 0;
            nzb = fs + 1;
            if (nzb == 5) {                                  /* ROS */
                uint32_t left_rsi, left_seg;
                b = (uint32_t)(c->filled / bs);
                left_rsi = c->p->rsi - b;
                left_seg = 64 - (b % 64);
                nzb = left_rsi < left_seg ? left_rsi : left_seg;
            } else if (nzb > 5) {
                nzb--;
            }
            count = nzb * bs - (uint32_t)ref;
            if ((size_t)c->p->rsi * bs - c->filled < count)
                return AECO_DATA_ERROR;                      /* decode.c:543-544 */
            for (i = 0; i < count; i++)
                if (!dec_emit(c, 0))
                    return 0;
            return 1;
        }
    }

    if (id == (UINT32_C(1) << c->dv.id_len) - 1) {           /* decode.c:646-677 */
        for (i = 0; i < bs; i++)
            if (!dec_copy_sample(c))
                return 0;
        return 1;
    }

    {                                                        /* decode.c:423-502 split */
        int k = (int)id - 1;
        uint32_t n = bs - (uint32_t)ref;
        uint32_t fsv[4096];
        uint32_t *fsp = fsv, *heap = NULL;
        int ok = 1;

        if (c->dv.id_len <= 1)
            return AECO_DATA_ERROR;   /* cannot happen: id is 0 or all-ones */
        if (ref && !dec_copy_sample(c))
            return 0;
        if (n > 4096) {
            heap = (uint32_t *)malloc(n * sizeof *heap);
            if (!heap)
                return AECO_MEM_ERROR;
            fsp = heap;
        }
        for (i = 0; i < n && ok; i++)                        /* all FS parts first */
            ok = br_unary(&c->r, &fsp[i]);
        for (i = 0; i < n && ok; i++) {
            if (!br_has(&c->r, k) || c->budget == 0) {
                ok = 0;
                break;
            }
            dec_emit(c, (fsp[i] << k) + (k ? br_get(&c->r, k) : 0));
        }
        free(heap);
        return ok;
    }
}

int aeco_decode(const aeco_params *p, const uint8_t *in, size_t in_len,
                uint8_t *out, size_t out_cap, size_t *out_len, uint64_t *in_used)
{
    dec_ctx c;
    int rc;
    uint64_t good = 0;

    memset(&c, 0, sizeof c);
    rc = aeco_derive(p, 0, &c.dv);
    if (rc != AECO_OK)
        return rc;
    c.p = p;
    c.r.in = in;
    c.r.nbits = (uint64_t)in_len * 8;
    c.out = out;
    c.budget = out_cap / (size_t)c.dv.bytes_per_sample;
    c.buf = (uint32_t *)malloc((size_t)p->rsi * p->block_size * sizeof *c.buf);
    if (!c.buf)
        return AECO_MEM_ERROR;

    for (;;) {
        rc = dec_cds(&c);
        if (rc <= 0)
            break;
        good = c.r.pos;
    }
    if (rc == 0) {
        size_t left = out_cap - (c.written + (c.filled - c.flushed))
                                * (size_t)c.dv.bytes_per_sample;
        if (left > 0 && left < (size_t)c.dv.bytes_per_sample)
            rc = AECO_MEM_ERROR;                             /* decode.c:821-823 */
        else
            dec_flush(&c);                                   /* decode.c:825 */
    }
    free(c.buf);
    if (out_len)
        *out_len = c.written * (size_t)c.dv.bytes_per_sample;
    if (in_used)
        *in_used = good;
    return rc;
}
