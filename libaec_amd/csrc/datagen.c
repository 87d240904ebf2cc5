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
