// aec_lane.h -- per-lane building blocks of the MI355X adaptive-entropy codec.
//
// Everything here is the work ONE lane does on ONE block (encode) or ONE coded data set
// (decode).  The functions are __host__ __device__ so the same arithmetic can be exercised
// by the CPU unit tests (tests/emul) without a GPU; the product only ever calls them from
// the HIP kernels in aec_enc.hip / aec_dec.hip.
//
// Reformulation of the reference's sequential k search (reference src/encode.c:329-410):
// len(k) = fs(k) + n*(k+1) is convex in k because g(k) = fs(k) - fs(k+1) never increases, so
// its minimum is attained on one contiguous plateau [klo, khi].  The reference's hill climb
// started at the previous block's k returns clamp(k_prev, klo, khi) and always the same
// minimal length.  A block is therefore summarised by (len_min, klo, khi); the carried k is
// a composition of clamps, which is associative and is resolved by a scan (aec_enc.hip).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AEC_HD __host__ __device__ __forceinline__
#else
#define AEC_HD inline
#endif

// blocks of up to this many samples read all their codes from one 64-bit peek (decode_block_noref)
#ifndef AEC_DEC_GRP
#define AEC_DEC_GRP 16
#endif

// AEC_ANY(x): wave-uniform "some lane needs this" test used to enter rare paths once per
// wavefront on the device; on the host (one lane at a time) it is just x.
#if defined(__HIP_DEVICE_COMPILE__)
#define AEC_ANY(x) __any(x)
#else
#define AEC_ANY(x) (x)
#endif

namespace aec {

// ------------------------------------------------------------------------------------
// stream parameters as the kernels see them
// ------------------------------------------------------------------------------------
enum : uint32_t {
    F_SIGNED = 1, F_3BYTE = 2, F_MSB = 4, F_PREPROCESS = 8, F_RESTRICTED = 16, F_PAD_RSI = 32,
    F_NOT_ENFORCE = 64
};

struct Cfg {
    uint32_t bps;            // bits per sample 1..32
    uint32_t bs;             // block size in samples (even, <= 64)
    uint32_t rsi;            // blocks per reference sample interval (1..4096)
    uint32_t flags;
    uint32_t id_len;         // 1..5   (reference encode.c:804-859)
    uint32_t bytes;          // container bytes per sample 1..4
    uint32_t kmax;           // (1 << id_len) - 3 (encode.c:872); 0 when id_len == 1
    uint32_t xmin, xmax;     // encode.c:862-870
    uint32_t segs_per_rsi;   // ceil(rsi / 64): a segment = 64 blocks = one wavefront pass
    uint32_t pad0;
    uint64_t total_samples;  // samples in this batch
    uint64_t total_blocks;   // ceil(total_samples / bs)  (encode.c:676-684 pads the last one)
    uint64_t total_segs;
    uint64_t rsi_count;      // ceil(total_blocks / rsi)
};

enum : uint32_t { OPT_ZERO = 0, OPT_SE = 1, OPT_SPLIT = 2, OPT_UNCOMP = 3, OPT_ZCONT = 4 };

// Per-block summary written by the analysis kernel and consumed by the packing kernel.
//   [0:12)  CDS length in bits (<= 5 + 32 + 64*32 = 2085)
//   [12:15) option
//   [16:21) klo   (split plateau, or for OPT_ZERO the run's unary value, 7 bits [16:23))
//   [24:29) khi   (in HBM the kernels store, for a block that updates k, the composition of the clamps of its
//                  segment's blocks up to and including it instead of its own plateau: aec_enc.hip analyze_segment)
AEC_HD uint32_t meta_pack(uint32_t len, uint32_t opt, uint32_t a, uint32_t b)
{
    return len | (opt << 12) | (a << 16) | (b << 24);
}
AEC_HD uint32_t meta_len(uint32_t m) { return m & 0xFFFu; }
AEC_HD uint32_t meta_opt(uint32_t m) { return (m >> 12) & 7u; }
AEC_HD uint32_t meta_a(uint32_t m) { return (m >> 16) & 0xFFu; }
AEC_HD uint32_t meta_b(uint32_t m) { return (m >> 24) & 0xFFu; }

// k transfer function of a run of blocks: k_out = min(max(k_in, lo), hi)
struct KClamp {
    uint32_t lo, hi;
};
AEC_HD KClamp kclamp_identity() { return KClamp{0u, 31u}; }
AEC_HD uint32_t kclamp_apply(KClamp f, uint32_t k) { return k < f.lo ? f.lo : (k > f.hi ? f.hi : k); }
// first f, then g
AEC_HD KClamp kclamp_then(KClamp f, KClamp g)
{
    return KClamp{kclamp_apply(g, f.lo), kclamp_apply(g, f.hi)};
}

AEC_HD uint32_t low_mask32(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

AEC_HD uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

AEC_HD int bit_length64(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }

// ------------------------------------------------------------------------------------
// sample access and the preprocessor (reference encode_accessors.c:61-143, encode.c:235-311)
// ------------------------------------------------------------------------------------
AEC_HD uint32_t load_sample_bytes(const uint8_t *p, uint32_t bytes, bool msb)
{
    uint32_t v = 0;
    if (msb) {
        for (uint32_t i = 0; i < bytes; i++) v = (v << 8) | p[i];
    } else {
        for (uint32_t i = bytes; i-- > 0;) v = (v << 8) | p[i];
    }
    return v;
}

AEC_HD uint32_t sign_extend(uint32_t v, uint32_t bps)
{
    const uint32_t m = 1u << (bps - 1);
    return (v ^ m) - m;
}

// Mapped prediction error of `cur` given its predecessor `prev` (unit-delay predictor).
// Both flavours are written as selects (no control flow) so that the 64 lanes of a wavefront,
// which see different signs of the difference, stay on one instruction stream.
// Unsigned flavour: reference encode.c:255-269.
AEC_HD uint32_t pp_unsigned(uint32_t prev, uint32_t cur, uint32_t xmax)
{
    const bool up = cur >= prev;
    const uint32_t delta = up ? cur - prev : prev - cur;
    const uint32_t room = up ? prev : xmax - prev;          // how far the mapping stays two-sided
    const uint32_t folded = 2u * delta - (up ? 0u : 1u);
    const uint32_t clipped = up ? cur : xmax - cur;
    return delta <= room ? folded : clipped;
}

// Signed flavour: reference encode.c:294-309; inputs already sign extended, arithmetic modulo 2^32.
AEC_HD uint32_t pp_signed(uint32_t prev_u, uint32_t cur_u, uint32_t xmin, uint32_t xmax)
{
    const bool down = (int32_t)cur_u < (int32_t)prev_u;
    const uint32_t delta = down ? prev_u - cur_u : cur_u - prev_u;
    const uint32_t room = down ? xmax - prev_u : prev_u - xmin;
    const uint32_t folded = 2u * delta - (down ? 1u : 0u);
    const uint32_t clipped = down ? xmax - cur_u : cur_u - xmin;
    return delta <= room ? folded : clipped;
}

AEC_HD uint32_t pp_any(uint32_t prev_raw, uint32_t cur_raw, const Cfg &c)
{
    if (c.flags & F_SIGNED)
        return pp_signed(sign_extend(prev_raw, c.bps), sign_extend(cur_raw, c.bps), c.xmin, c.xmax);
    return pp_unsigned(prev_raw, cur_raw, c.xmax);
}

// ------------------------------------------------------------------------------------
// block analysis
// ------------------------------------------------------------------------------------
template <int BS, bool WIDE>
AEC_HD uint64_t fs_sum(const uint32_t *d, uint32_t bs_rt, uint32_t k)
{
    const uint32_t bs = BS ? (uint32_t)BS : bs_rt;
    if (WIDE) {
        uint64_t s = 0;
#pragma unroll
        for (uint32_t i = 0; i < bs; i++) s += (uint64_t)(d[i] >> k);
        return s;
    } else {
        uint32_t s = 0;   // bps <= 16, bs <= 64: at most 64 * 65535 < 2^22
#pragma unroll
        for (uint32_t i = 0; i < bs; i++) s += d[i] >> k;
        return s;
    }
}

// Split option: plateau [klo, khi] of minimisers of len(k) over k in [0, kmax] and the minimal
// length (reference encode.c:329-410; see the note at the top of this file).
//   n = number of coded samples (bs - ref); the sum runs over the whole block because the
//   reference sample slot holds d = 0 (encode.c:254, 323-324).
// Plateau search of the split option over any evaluator fs(k) = sum of (sample >> k): the
// kernels' packed 16-bit evaluator and the plain one below share this control flow.
template <class FS>
AEC_HD void assess_split_with(FS fs, uint32_t n, uint32_t kmax, uint32_t &klo, uint32_t &khi,
                              uint32_t &len_min)
{
    using T = decltype(fs(0u));        // 32-bit sums where the evaluator guarantees they fit
    const T s0 = fs(0u);
    // smallest k that can possibly satisfy g(k) <= n needs n * 2^(k+2) >= s0
    int ks = bit_length64(s0) - bit_length64(n) - 2;
    if (ks < 0) ks = 0;
    if ((uint32_t)ks > kmax) ks = (int)kmax;

    uint32_t k = (uint32_t)ks;
    T f_cur = k ? fs(k) : s0;
    bool have_lo = false;
    klo = khi = kmax;
    len_min = 0;
    for (;;) {
        if (k >= kmax) {
            if (!have_lo) {
                klo = kmax;
                len_min = (uint32_t)(f_cur + (uint64_t)n * (k + 1));
            }
            khi = kmax;
            break;
        }
        const T f_next = fs(k + 1u);
        const T g = f_cur - f_next;
        if (!have_lo && g <= n) {
            have_lo = true;
            klo = k;
            len_min = (uint32_t)(f_cur + (uint64_t)n * (k + 1));
        }
        if (g < n) {
            khi = k;
            break;
        }
        k++;
        f_cur = f_next;
    }
}

template <int BS, bool WIDE>
AEC_HD void assess_split(const uint32_t *d, uint32_t bs_rt, uint32_t n, uint32_t kmax,
                         uint32_t &klo, uint32_t &khi, uint32_t &len_min)
{
    assess_split_with([&](uint32_t k) { return (uint64_t)fs_sum<BS, WIDE>(d, bs_rt, k); }, n, kmax, klo, khi,
                      len_min);
}

// Second-extension option length, exact replica of the uint64_t arithmetic and the in-order
// early exit of reference encode.c:412-434.
template <int BS>
AEC_HD uint32_t assess_se(const uint32_t *d, uint32_t bs_rt, uint32_t limit)
{
    const uint32_t bs = BS ? (uint32_t)BS : bs_rt;
    uint32_t len = 1;
    bool over = false;
#pragma unroll
    for (uint32_t i = 0; i < bs; i += 2) {
        const uint32_t a = d[i], b = d[i + 1];
        const uint32_t s = a + b;
        if (!over) {
            if (s < a) {
                // a + b >= 2^32: replicate the wrapping 64-bit product (encode.c:428-429)
                const uint64_t s64 = (uint64_t)a + (uint64_t)b;
                const uint64_t l64 = (uint64_t)len + s64 * (s64 + 1) / 2 + b + 1;
                if (l64 > limit) over = true; else len = (uint32_t)l64;
            } else if (s >= 32768u) {
                over = true;    // s(s+1)/2 >= 2^29 > any limit (limit <= 64*32)
            } else {
                len += s * (s + 1) / 2 + b + 1;
                if (len > limit) over = true;
            }
        }
    }
    return over ? 0xFFFFFFFFu : len;
}

struct BlockChoice {
    uint32_t opt;    // OPT_SE / OPT_SPLIT / OPT_UNCOMP
    uint32_t bits;   // whole CDS length: id + optional reference sample + payload
    uint32_t klo, khi;
};

// reference encode.c:585-612 m_select_code_option (plus the CDS framing of 520-563)
// the choice between the assessed options (reference encode.c:585-612, note the tie rules)
AEC_HD BlockChoice choose_from(const Cfg &c, uint32_t bs, uint32_t ref, uint32_t split_len, uint32_t se_len,
                               uint32_t klo, uint32_t khi)
{
    const uint32_t n = bs - ref;
    const uint32_t uncomp_len = n * c.bps;         // encode.c:270, 746: (bs - ref) * bps
    const uint32_t head = c.id_len + ref * c.bps;
    BlockChoice r;
    r.klo = klo;
    r.khi = khi;
    if (split_len < uncomp_len) {
        if (split_len < se_len) { r.opt = OPT_SPLIT; r.bits = head + split_len; }
        else                    { r.opt = OPT_SE;    r.bits = head + se_len; }
    } else {
        if (uncomp_len <= se_len) { r.opt = OPT_UNCOMP; r.bits = c.id_len + bs * c.bps; }
        else                      { r.opt = OPT_SE;     r.bits = head + se_len; }
    }
    return r;
}

template <int BS, bool WIDE>
AEC_HD BlockChoice choose_option(const uint32_t *d, const Cfg &c, uint32_t ref)
{
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const uint32_t n = bs - ref;
    uint32_t split_len = 0xFFFFFFFFu, klo = 0, khi = 31;
    if (c.id_len > 1)
        assess_split<BS, WIDE>(d, bs, n, c.kmax, klo, khi, split_len);
    const uint32_t se_len = assess_se<BS>(d, bs, n * c.bps);
    return choose_from(c, bs, ref, split_len, se_len, klo, khi);
}

// Zero-block run bookkeeping inside one segment (reference encode.c:614-659, 565-583).
//   zmask   bit l set <=> block l of the segment is all zero (only bits < nv may be set)
//   l       this block (must be a zero block)
//   nv      blocks in the segment (the segment ends at a 64-block boundary or at the RSI end)
// Returns 0 for a block swallowed by a run that started earlier, else the run length; `fs`
// receives the unary value of the run's CDS.
AEC_HD uint32_t zero_run_at(uint64_t zmask, uint32_t l, uint32_t nv, uint32_t &fs)
{
    if (l > 0 && ((zmask >> (l - 1)) & 1)) return 0;
    const uint64_t rest = ~(zmask >> l);                  // first 0 bit = end of run
    uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
    if (run > nv - l) run = nv - l;
    const bool closes = (l + run == nv);                  // encode.c:649
    if (closes && run > 4) fs = 4;                        // ROS, encode.c:650-651, 574-575
    else if (run >= 5)     fs = run;                      // encode.c:576-577
    else                   fs = run - 1;                  // encode.c:578-579
    return run;
}

// ------------------------------------------------------------------------------------
// bit emission: MSB-first writer over a zero-initialised array of 32-bit words whose
// bit 31 is the earliest stream bit.  Sink::or_word(idx, value) ORs into word idx
// (an LDS atomic on the device, a plain |= in the host tests).
// ------------------------------------------------------------------------------------
template <class Sink>
struct BitWriter {
    Sink &sink;
    uint64_t acc;
    uint32_t n;      // pending bits in acc (low n bits), < 32 between calls
    uint32_t word;   // next word to be written

    AEC_HD BitWriter(Sink &s, uint32_t bit_offset)
        : sink(s), acc(0), n(bit_offset & 31u), word(bit_offset >> 5) {}

    AEC_HD void put(uint32_t v, uint32_t nb)   // nb <= 32, v < 2^nb
    {
        acc = (acc << nb) | v;
        n += nb;
        if (n >= 32) {
            n -= 32;
            sink.or_word(word++, (uint32_t)(acc >> n));
        }
    }
    AEC_HD void unary(uint32_t zeros)          // `zeros` 0 bits then a 1 (encode.c:85-104)
    {
        if (zeros < 32) {
            put(1u, zeros + 1u);
        } else {
            if (n) sink.or_word(word, (uint32_t)(acc << (32 - n)));
            const uint32_t pos = n + zeros;
            word += pos >> 5;
            n = pos & 31u;
            acc = 0;
            put(1u, 1u);
        }
    }
    AEC_HD void finish()
    {
        if (n) sink.or_word(word, (uint32_t)(acc << (32 - n)));
    }
};

// Split option for blocks of 32 or 64 samples (reference encode.c:520-534): the unary parts are
// assembled sixteen at a time and the fields eight at a time in a 64-bit register and appended with
// two puts per group, instead of one put (and its flush test) per sample.  A group that does not
// fit -- unary parts of more than 64 bits, k above 8 -- is appended sample by sample.
template <int BS, class Sink>
AEC_HD void emit_split_groups(BitWriter<Sink> &w, const uint32_t *d, const Cfg &c, uint32_t k, uint32_t ref,
                              uint32_t ref_sample, bool live)
{
    static_assert(BS % 16 == 0, "whole groups");
    if (!live) return;
    w.put(k + 1u, c.id_len);
    if (ref) w.put(ref_sample, c.bps);
#pragma unroll
    for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += 16) {
        uint64_t ua = 0;
        uint32_t bits = 0;
        bool big = false;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t i = g0 + j;
            const uint32_t t = d[i] >> k;
            const bool skip = i == 0 && ref;                // the reference slot carries no code
            big = big || (!skip && t >= 64u);
            const uint32_t n1 = skip ? 0u : t + 1u;
            bits += n1 & 127u;
            ua = (ua << (n1 & 63u)) | (skip ? 0u : 1u);
        }
        if (!big && bits <= 64u) {
            if (bits > 32u) w.put((uint32_t)(ua >> 32), bits - 32u);
            if (bits) w.put((uint32_t)ua, bits > 32u ? 32u : bits);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; j++)
                if (!(g0 + j == 0 && ref)) w.unary(d[g0 + j] >> k);
        }
    }
    if (k != 0 && k <= 8u) {
        const uint32_t m = low_mask32(k);
#pragma unroll
        for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += 8) {
            uint64_t fa = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const bool skip = g0 + j == 0 && ref;
                fa = (fa << (skip ? 0u : k)) | (skip ? 0u : d[g0 + j] & m);
            }
            const uint32_t bits = (8u - ((g0 == 0 && ref) ? 1u : 0u)) * k;
            if (bits > 32u) w.put((uint32_t)(fa >> 32), bits - 32u);
            w.put((uint32_t)fa, bits > 32u ? 32u : bits);
        }
    } else if (k != 0 && k <= 16u) {
        // fields of 9..16 bits: two per put
        const uint32_t m = low_mask32(k);
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BS; i += 2) {
            const bool skip = i == 0 && ref;
            const uint32_t a = skip ? 0u : d[i] & m;
            w.put((a << k) | (d[i + 1] & m), skip ? k : 2u * k);
        }
    } else if (k != 0) {
        const uint32_t m = low_mask32(k);
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BS; i++)
            if (i >= ref) w.put(d[i] & m, k);
    }
    w.finish();
}

// Emit one block's CDS (reference encode.c:520-583).  d[0] of a reference block is ignored
// for SPLIT/SE payloads exactly as the reference does (sample slot 0 holds 0 / is skipped).
template <int BS, class Sink>
AEC_HD void emit_block(BitWriter<Sink> &w, const uint32_t *d, const Cfg &c, uint32_t opt,
                       uint32_t k_or_fs, uint32_t ref, uint32_t ref_sample)
{
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    if (opt == OPT_SPLIT) {
        const uint32_t k = k_or_fs;
        w.put(k + 1u, c.id_len);
        if (ref) w.put(ref_sample, c.bps);
        if (BS != 0 && BS % 4 == 0) {
            // fundamental sequences four at a time: one put while the four codes fit 32 bits
#pragma unroll
            for (uint32_t i = 0; i < bs; i += 4) {
                uint32_t v = 0, bits = 0;
                bool wide = false;
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const bool skip = i + j == 0 && ref;            // the reference slot carries no code
                    const uint32_t t = d[i + j] >> k;
                    wide = wide || (!skip && t >= 8u);
                    const uint32_t n1 = skip ? 0u : (t & 7u) + 1u;
                    v = (v << n1) | (skip ? 0u : 1u);
                    bits += n1;
                }
                if (!wide) {
                    if (bits) w.put(v, bits);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++)
                        if (!(i + j == 0 && ref)) w.unary(d[i + j] >> k);
                }
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < bs; i++)
                if (i >= ref) w.unary(d[i] >> k);
        }
        if (k != 0 && k <= 16u && BS != 0) {
            // fields of up to 16 bits: two per put
            const uint32_t m = low_mask32(k);
#pragma unroll
            for (uint32_t i = 0; i < bs; i += 2) {
                const bool skip = i == 0 && ref;
                const uint32_t a = skip ? 0u : d[i] & m;
                w.put((a << k) | (d[i + 1] & m), skip ? k : 2u * k);
            }
        } else if (k) {
            const uint32_t m = low_mask32(k);
#pragma unroll
            for (uint32_t i = 0; i < bs; i++)
                if (i >= ref) w.put(d[i] & m, k);
        }
    } else if (opt == OPT_SE) {
        w.put(1u, c.id_len + 1u);
        if (ref) w.put(ref_sample, c.bps);
#pragma unroll
        for (uint32_t i = 0; i < bs; i += 2) {
            const uint32_t s = d[i] + d[i + 1];
            w.unary(s * (s + 1u) / 2u + d[i + 1]);
        }
    } else if (opt == OPT_UNCOMP) {
        const uint32_t m = low_mask32(c.bps);
        w.put((1u << c.id_len) - 1u, c.id_len);
        if (BS != 0 && c.bps <= 16u) {
            // samples of up to 16 bits: two per put
#pragma unroll
            for (uint32_t i = 0; i < bs; i += 2) {
                const uint32_t a = (i == 0 && ref) ? ref_sample : (d[i] & m);
                w.put((a << c.bps) | (d[i + 1] & m), 2u * c.bps);
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < bs; i++)
                w.put((i == 0 && ref) ? ref_sample : (d[i] & m), c.bps);
        }
    } else if (opt == OPT_ZERO) {
        w.put(0u, c.id_len + 1u);
        if (ref) w.put(ref_sample, c.bps);
        w.unary(k_or_fs);
    }
    w.finish();
}

// Fast emission for small blocks (BS <= 16): when the unary region (fundamental sequences /
// second-extension codes / the zero-run code) and the k-bit field region of a CDS each fit in
// 64 bits, both are assembled in registers with straight-line code and leave through at most
// four word puts, instead of two bit-writer calls per sample.  For the split option at its
// optimal k the unary region is at most 3*n bits (fs(k) <= 2n follows from g(k) <= n), so with
// n <= 16 only k = kmax blocks can miss the fast path.  Returns the eligibility test so the
// caller can route the remaining lanes through emit_block().
AEC_HD bool small_eligible(const Cfg &c, uint32_t bs, uint32_t opt, uint32_t k, uint32_t ref, uint32_t len,
                           uint32_t &ubits, uint32_t &fbits)
{
    const uint32_t n = bs - ref;
    const uint32_t head = c.id_len + (opt == OPT_SPLIT ? 0u : 1u) + ref * c.bps;
    fbits = opt == OPT_SPLIT ? n * k : 0u;
    ubits = len - head - fbits;
    return (opt == OPT_SPLIT || opt == OPT_SE || opt == OPT_ZERO) && ubits <= 64 && fbits <= 64 && len >= head;
}

template <int BS, class Sink>
AEC_HD void emit_small(BitWriter<Sink> &w, const uint32_t *d, const Cfg &c, uint32_t opt, uint32_t k_or_fs,
                       uint32_t ref, uint32_t ref_sample, uint32_t ubits, uint32_t fbits, bool live)
{
    const bool is_split = live && opt == OPT_SPLIT, is_se = live && opt == OPT_SE;
    const bool is_zero = live && opt == OPT_ZERO;
    const uint32_t k = is_split ? k_or_fs : 0u;
    const uint32_t km = low_mask32(k);
    // Unary and field regions, right aligned.  The loop carries no per-sample predicate: lanes
    // that are not split blocks run it on k = 0 and have both registers replaced afterwards; only
    // sample 0 can be the reference slot (skipped).  Shifts are taken modulo 64: a lane whose
    // codes do not fit is not `live` here (small_eligible) and goes through emit_block.
    uint64_t ua = 0, fa = 0;
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)BS; i++) {
        const uint32_t t = d[i] >> k;
        if (i == 0) {
            ua = ref ? 0u : 1u;                          // (1 << (t + 1)) >> (t + 1) == the lone 1
            fa = ref ? 0u : (uint64_t)(d[0] & km);
        } else {
            ua = (ua << ((t + 1u) & 63u)) | 1u;
            fa = (fa << k) | (d[i] & km);
        }
    }
    if (!is_split) {
        ua = is_zero ? 1u : 0u;
        fa = 0;
    }
    if (AEC_ANY(is_se)) {
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BS; i += 2) {
            const uint32_t sum = d[i] + d[i + 1];
            const uint32_t m = sum * (sum + 1u) / 2u + d[i + 1];
            ua = (ua << (is_se ? (m + 1u) & 63u : 0u)) | (is_se ? 1u : 0u);
        }
    }
    // header: option id (+ the low-entropy selector bit), then the reference sample; it rides in
    // front of the unary region when the two fit one 64-bit register -- one put less
    const uint32_t idv = is_split ? k + 1u : (is_se ? 1u : 0u);
    const uint32_t idb = c.id_len + (is_split ? 0u : 1u);
    const uint32_t hb = idb + (ref ? c.bps : 0u);
    const bool merged = hb + ubits <= 64u;
    if (AEC_ANY(live && !merged)) {
        if (live && !merged) {
            w.put(idv, idb);
            if (ref) w.put(ref_sample, c.bps);
        }
    }
    if (live) {
        uint32_t nb = ubits;
        if (merged) {
            const uint64_t hdr = ref ? ((uint64_t)idv << c.bps) | ref_sample : (uint64_t)idv;
            ua |= hdr << (ubits & 63u);                  // merged implies ubits < 64 (hb >= 1)
            nb = ubits + hb;
        }
        if (nb > 32) w.put((uint32_t)(ua >> 32), nb - 32);
        w.put((uint32_t)ua, nb > 32 ? 32u : nb);
        if (fbits > 32) w.put((uint32_t)(fa >> 32), fbits - 32);
        if (fbits) w.put((uint32_t)fa, fbits > 32 ? 32u : fbits);
        w.finish();
    }
}

// ------------------------------------------------------------------------------------
// bit reader: stream = big-endian 32-bit words at a 4-byte aligned base
// ------------------------------------------------------------------------------------
// Fetch policy of BitReaderT: Fetch::operator()(idx) returns stream word idx in host order, 0 past
// the end.  MemFetch reads the stream where it lies; the RSI index kernel substitutes a fetcher
// that serves words from an LDS window refilled by the whole wavefront.
struct MemFetch {
    const uint32_t *words;
    uint64_t nwords;
    AEC_HD uint32_t operator()(uint64_t idx) const { return idx < nwords ? bswap32(words[idx]) : 0u; }
};

template <class Fetch>
struct BitReaderT {
    Fetch f;
    uint64_t end_bit;    // bits that belong to the stream (reads beyond it see zeros)
    uint64_t pos;        // absolute position of the next unread bit
    uint64_t win;        // unread bits, left aligned; bits below the top `cnt` are zero
    uint32_t cnt;        // valid bits in win; (pos + cnt) is always a multiple of 32
    uint64_t next_word;

    AEC_HD uint32_t fetch(uint64_t idx) const { return f(idx); }

    AEC_HD void init(const uint32_t *w, uint64_t nw, uint64_t endb, uint64_t start_bit)
    {
        init(Fetch{w, nw}, endb, start_bit);
    }
    AEC_HD void init(const Fetch &fetcher, uint64_t endb, uint64_t start_bit)
    {
        f = fetcher; end_bit = endb; pos = start_bit;
        const uint32_t sh = (uint32_t)(start_bit & 31u);
        next_word = (start_bit >> 5) + 1;
        win = (uint64_t)fetch(start_bit >> 5) << (32 + sh);
        cnt = 32 - sh;
    }
    AEC_HD void refill()   // requires cnt <= 32
    {
        win |= (uint64_t)fetch(next_word++) << (32 - cnt);
        cnt += 32;
    }
    AEC_HD uint32_t get(uint32_t n)   // n <= 32
    {
        if (n == 0) return 0;
        if (cnt < n) refill();
        const uint32_t v = (uint32_t)(win >> (64 - n));
        win <<= n;
        cnt -= n;
        pos += n;
        return v;
    }
    AEC_HD void skip(uint64_t n)
    {
        if (n <= cnt) {
            win = n >= 64 ? 0 : win << n;
            cnt -= (uint32_t)n;
            pos += n;
        } else {
            init(f, end_bit, pos + n);
        }
    }
    // counts zeros up to the next 1 (reference decode.c:288-340); false if the stream ends first
    AEC_HD bool unary(uint32_t &zeros)
    {
        uint32_t z = 0;
        for (;;) {
            if (win != 0) {
                const uint32_t c = (uint32_t)__builtin_clzll(win);
                win = c >= 63 ? 0 : win << (c + 1);
                cnt -= c + 1;
                pos += c + 1;
                zeros = z + c;
                return true;
            }
            z += cnt;
            pos += cnt;
            cnt = 0;
            if (pos >= end_bit) return false;
            refill();
        }
    }
    // skips `n` unary codes; false if the stream ends first
    AEC_HD bool skip_unary(uint32_t n)
    {
        while (n) {
            if (cnt == 0) {
                if (pos >= end_bit) return false;
                refill();
            }
            const uint32_t ones = (uint32_t)__builtin_popcountll(win);
            if (ones < n) {
                n -= ones;
                pos += cnt;
                cnt = 0;
                win = 0;
            } else {
                uint32_t z;
                for (; n; n--) unary(z);
            }
        }
        return true;
    }
    AEC_HD bool overrun() const { return pos > end_bit; }
};
using BitReader = BitReaderT<MemFetch>;

// Lean window reader used by the RSI-parallel decoder.  Src::word(i) returns the i-th 32-bit
// big-endian word (already in host order) counted from a per-lane base, or 0 past what is
// available (Src then remembers that it starved).  Between operations at least 32 unread bits sit
// in the window, so a field of up to 32 bits or a short unary code never needs a refill first.
template <class Src>
struct WinReader {
    Src src;
    uint64_t win;       // unread bits, left aligned; bits below the top `cnt` are zero
    uint32_t cnt;       // valid bits in win (32..64 between operations)
    uint32_t next;      // index (relative to the lane base) of the word held in `nextw`
    uint32_t nextw;     // word `next`, fetched ahead so that a refill is pure register work
    uint64_t base_bit;  // absolute bit position of relative word 0
    uint64_t end_bit;   // bits that belong to the stream

    AEC_HD void init(const Src &s, uint64_t base, uint64_t endb, uint32_t start_rel_bit)
    {
        src = s; base_bit = base; end_bit = endb;
        next = start_rel_bit >> 5;
        const uint32_t sh = start_rel_bit & 31u;
        const uint32_t w0 = src.word(next), w1 = src.word(next + 1);
        next += 2;
        nextw = src.word(next);
        win = (((uint64_t)w0 << 32) | w1) << sh;
        cnt = 64 - sh;
    }
    AEC_HD uint64_t pos() const { return base_bit + (uint64_t)next * 32u - cnt; }
    // Restores cnt >= 32 after at most 32 bits were consumed.  Written without control flow: the
    // appended word is masked out when no refill is due, and the look-ahead word is re-read from
    // the (LDS) source unconditionally.
    AEC_HD void top()
    {
        const bool take = cnt < 32;
        const uint64_t add = (uint64_t)nextw << ((32u - cnt) & 63u);
        win |= take ? add : 0;
        cnt += take ? 32u : 0u;
        next += take ? 1u : 0u;
        nextw = src.word(next);
    }
    AEC_HD uint32_t get(uint32_t n)   // 1 <= n <= 32
    {
        const uint32_t v = (uint32_t)(win >> (64 - n));
        win <<= n;
        cnt -= n;
        top();
        return v;
    }
    AEC_HD bool unary(uint32_t &zeros)
    {
        const uint32_t hi = (uint32_t)(win >> 32);
        if (__builtin_expect(hi != 0, 1)) {  // common case: the 1 bit is within 32 bits
            const uint32_t z = (uint32_t)__builtin_clz(hi);
            win <<= z + 1;
            cnt -= z + 1;
            top();
            zeros = z;
            return true;
        }
        uint32_t total = 0;
        for (;;) {
            if (win != 0) {
                const uint32_t c = (uint32_t)__builtin_clzll(win);
                win = c >= 63 ? 0 : win << (c + 1);
                cnt -= c + 1;
                if (cnt < 32) top();
                if (cnt < 32) top();
                zeros = total + c;
                return true;
            }
            total += cnt;
            cnt = 0;
            zeros = 0;
            if (pos() >= end_bit) return false;
            top();
        }
    }
    AEC_HD bool overrun() const { return pos() > end_bit || src.starved(); }
};

// Inverse of the preprocessor (reference decode.c:91-135), one step.
AEC_HD uint32_t unpp_unsigned(uint32_t x, uint32_t d, uint32_t xmax)
{
    const uint32_t med = xmax / 2 + 1;
    const uint32_t half = (d >> 1) + (d & 1);
    const uint32_t mask = (x & med) ? xmax : 0;
    if (half <= (mask ^ x)) return x + ((d & 1) ? ~(d >> 1) : (d >> 1));
    return mask ^ d;
}
AEC_HD uint32_t unpp_signed(uint32_t xu, uint32_t d, uint32_t xmax)
{
    const int32_t x = (int32_t)xu;
    const uint32_t half = (d >> 1) + (d & 1);
    const uint32_t step = (d & 1) ? ~(d >> 1) : (d >> 1);
    if (x < 0) {
        if (half <= xmax + xu + 1u) return xu + step;
        return d - xmax - 1u;
    }
    if (half <= xmax - xu) return xu + step;
    return xmax - d;
}

// second-extension code value m -> (a + b, m - tri(a + b)); reference decode.c:679-692 builds
// the same mapping as a 91-entry table.  false for m > 90 (outside the table).
AEC_HD bool se_lookup(uint32_t m, uint32_t &sum, uint32_t &second)
{
    // s = the largest s with s(s+1)/2 <= m, i.e. floor((sqrt(8m + 1) - 1) / 2).  8m + 1 <= 721 is exact in
    // float; at a triangular m the root is the integer 2s + 1 exactly, everywhere else it stays at least
    // 4 / 27 below the next odd integer, so a root good to a few ulp gives the exact floor (checked for
    // all 91 codes in tests/test_lane_emul.py) -- no loop, no table.
    const uint32_t mm = m > 90 ? 90u : m;
#if defined(__HIP_DEVICE_COMPILE__)
    const float root = __builtin_amdgcn_sqrtf((float)(8u * mm + 1u));
#else
    const float root = sqrtf((float)(8u * mm + 1u));
#endif
    const uint32_t s = (uint32_t)((root - 1.0f) * 0.5f);
    sum = s;
    second = mm - ((s * (s + 1u)) >> 1);
    return m <= 90;
}

// Decode status codes shared by the kernels and the host
enum : uint32_t { DEC_OK = 0, DEC_NEED_INPUT = 1, DEC_DATA_ERROR = 2 };
// bit 31 of a decode record's `pad`: a coded data set outgrew a lane's look-ahead and the batch was decoded again
// by the sequential path (aec_dec.hip: k_decode_redo); the low bits keep their meaning
constexpr uint32_t kDecRedo = 0x80000000u;

// Parses one CDS.  d receives the block's samples for SPLIT / SE / UNCOMP (d[0] is the
// reference sample on a reference block).  For a zero run `nzero_blocks` receives the number
// of blocks covered and d is untouched (only d[0] = reference sample if ref).
//   blk_in_rsi = index of this block inside its RSI (needed for ROS, decode.c:528-530)
template <int BS, class Reader>
AEC_HD uint32_t parse_cds(Reader &r, uint32_t *d, const Cfg &c, uint32_t ref,
                          uint32_t blk_in_rsi, uint32_t &nzero_blocks)
{
    // No early exits inside the sample loops: a lane that runs out of input keeps going on zeros
    // (every read is bounded by the reader) and reports once at the end, which keeps the 64 lanes
    // of a wavefront on one control path.
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    nzero_blocks = 0;
    bool short_input = false, corrupt = false;
    const uint32_t id = r.get(c.id_len);
    if (id == 0) {                                       // decode.c:634-644, 618-632
        const uint32_t sel = r.get(1);
        if (ref) d[0] = r.get(c.bps);
        if (sel) {                                       // decode.c:589-616
            uint32_t i = ref;
#pragma unroll
            for (uint32_t j = 0; j < bs / 2; j++) {
                uint32_t m = 0, s = 0, second = 0;
                if (!r.unary(m)) short_input = true;
                else if (!se_lookup(m, s, second)) corrupt = true;
                if ((i & 1u) == 0) d[i++] = s - second;
                d[i++] = second;
            }
        } else {                                         // decode.c:518-558
            uint32_t fs = 0;
            if (!r.unary(fs)) short_input = true;
            uint32_t nz = fs + 1;
            if (nz == 5) {
                const uint32_t left_rsi = c.rsi - blk_in_rsi;
                const uint32_t left_seg = 64 - (blk_in_rsi % 64);
                nz = left_rsi < left_seg ? left_rsi : left_seg;
            } else if (nz > 5) {
                nz--;
            }
            if (nz > c.rsi - blk_in_rsi) corrupt = true;   // decode.c:543-544
            nzero_blocks = nz;
        }
    } else if (id == (1u << c.id_len) - 1u) {            // decode.c:659-677
#pragma unroll
        for (uint32_t i = 0; i < bs; i++) d[i] = r.get(c.bps);
    } else {                                             // decode.c:462-502
        const uint32_t k = id - 1;
        if (ref) d[0] = r.get(c.bps);
#pragma unroll
        for (uint32_t i = 0; i < bs; i++)
            if (i >= ref) {
                uint32_t z = 0;
                if (!r.unary(z)) short_input = true;
                d[i] = z << k;
            }
        if (k) {
#pragma unroll
            for (uint32_t i = 0; i < bs; i++)
                if (i >= ref) d[i] += r.get(k);
        }
    }
    if (r.overrun() || short_input) return DEC_NEED_INPUT;
    return corrupt ? DEC_DATA_ERROR : DEC_OK;
}

// ------------------------------------------------------------------------------------
// Phase-structured block decoder (fast path of the RSI-parallel decode kernel).
//
// Instead of one sequential reader it works on a bit position `p` and a word source that can
// return ANY two consecutive words cheaply (an LDS ring on the device):
//   1. header        id, low-entropy selector, reference sample
//   2. unary phase   all fundamental sequences of the block, two codes per 32-bit peek; codes
//                    longer than 15 zeros take a (rare) slow step, entered wave-uniformly
//   3. field phase   the k-bit (or bps-bit) fields are at p + i*k: independent direct reads
//   4. combine       d[i] = (fs << k) + field; second extension / zero run fixed up last
// All code options run through the same phases with per-lane counts, so a wavefront whose lanes
// hold different options does not serialise.  Src::word2(i, w0, w1) yields words i and i+1.
// AEC_ANY(x) is __any(x) on the device (wave-uniform entry to the rare paths) and x on the host.
// ------------------------------------------------------------------------------------
template <class Src>
AEC_HD uint32_t peek32(Src &src, uint32_t p)
{
    uint32_t w0, w1;
    src.word2(p >> 5, w0, w1);
    return (uint32_t)((((((uint64_t)w0) << 32) | w1) << (p & 31u)) >> 32);
}

// 64 stream bits starting at bit position p (Src::word3 yields three consecutive words)
template <class Src>
AEC_HD uint64_t peek64(Src &src, uint32_t p)
{
    uint32_t w0, w1, w2;
    src.word3(p >> 5, w0, w1, w2);
    const uint32_t sh = p & 31u;
    const uint32_t hi = (uint32_t)((((((uint64_t)w0) << 32) | w1) << sh) >> 32);
    const uint32_t lo = (uint32_t)((((((uint64_t)w1) << 32) | w2) << sh) >> 32);
    return (((uint64_t)hi) << 32) | lo;
}

AEC_HD uint32_t clz32_or32(uint32_t v) { return v ? (uint32_t)__builtin_clz(v) : 32u; }

// general unary read at p for codes of any length; stops at end_p (stream end)
template <class Src>
AEC_HD uint32_t unary_slow(Src &src, uint32_t &p, uint32_t end_p, bool &short_input)
{
    uint32_t zeros = 0;
    for (;;) {
        const uint32_t h = peek32(src, p);
        if (h != 0) {
            const uint32_t z = (uint32_t)__builtin_clz(h);
            p += z + 1;
            return zeros + z;
        }
        zeros += 32;
        p += 32;
        if (p >= end_p) {
            short_input = true;
            return 0;
        }
    }
}

template <int BS, class Src>
AEC_HD uint32_t decode_block_any(Src &src, uint32_t &p, uint32_t end_p, uint32_t *d, const Cfg &c,
                             uint32_t ref, uint32_t blk_in_rsi, bool live, uint32_t &nzero_blocks)
{
    static_assert(BS >= 2 && BS % 2 == 0, "templated block sizes only");
    const uint32_t idmax = (1u << c.id_len) - 1u;
    bool short_input = false, corrupt = false;
    nzero_blocks = 0;

    // ---- 1. header --------------------------------------------------------------------------
    uint32_t h = peek32(src, p);
    const uint32_t id = h >> (32 - c.id_len);
    const bool lowent = live && id == 0;
    const bool unc = live && id == idmax;
    const bool split = live && !lowent && !unc;
    const uint32_t sel = (h >> (31 - c.id_len)) & 1u;
    const bool se = lowent && sel;
    const bool zero = lowent && !sel;
    const uint32_t k = split ? id - 1u : 0u;
    p += live ? c.id_len + (lowent ? 1u : 0u) : 0u;
    uint32_t refv = 0;
    if (AEC_ANY(ref != 0)) {                             // first block of an RSI (per lane)
        refv = peek32(src, p) >> (32 - c.bps);
        p += (live && ref && !unc) ? c.bps : 0u;
    }

    // ---- 2. unary phase: code for sample slot i lives in u[i] ---------------------------------
    // Eight slots per 64-bit peek: each code is located with one clz on the upper half and one
    // 64-bit shift.  (For the split option at its optimal k the whole unary region of a block is
    // at most 3*n bits, see emit_small.)  A group that does not fit 64 bits, or holds a code with
    // 32 or more zeros, is redone code by code -- entered wave-uniformly, practically never.
    const uint32_t nfs = split ? (uint32_t)BS - ref : (se ? (uint32_t)BS / 2 : (zero ? 1u : 0u));
    uint32_t u[BS];
    constexpr uint32_t GRP = BS < 8 ? (uint32_t)BS : 8u;
#pragma unroll
    for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += GRP) {
        uint64_t U = peek64(src, p);
        uint32_t used = 0;
        bool bad = false;
#pragma unroll
        for (uint32_t j = 0; j < GRP; j++) {
            const uint32_t i = g0 + j;
            const bool act = i >= ref && i - ref < nfs;
            const uint32_t z = clz32_or32((uint32_t)(U >> 32));
            bad = bad || (act && z >= 32);
            const uint32_t n1 = act ? z + 1u : 0u;
            U <<= (n1 & 63u);
            used += n1;
            u[i] = act ? z : 0u;
        }
        if (AEC_ANY(bad)) {
            if (bad) {
                uint32_t q = p;
#pragma unroll
                for (uint32_t j = 0; j < GRP; j++) {
                    const uint32_t i = g0 + j;
                    if (i >= ref && i - ref < nfs) u[i] = unary_slow(src, q, end_p, short_input);
                }
                used = q - p;
            }
        }
        p += used;
    }

    // ---- 3. field phase ---------------------------------------------------------------------
    const uint32_t kk = split ? k : (unc ? c.bps : 0u);
    const uint32_t nf = split ? (uint32_t)BS - ref : (unc ? (uint32_t)BS : 0u);
    const uint32_t off = split ? ref : 0u;
    if (AEC_ANY(kk > 16)) {
        // wide fields (uncompressed blocks, large k): one 32-bit peek per sample
        const uint32_t fsh = (32u - kk) & 31u;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BS; i++) {
            const bool act = kk != 0 && i >= off && i - off < nf;
            const uint32_t v = peek32(src, act ? p + (i - off) * kk : p);
            const uint32_t f = act ? v >> fsh : 0u;
            d[i] = (u[i] << k) + f;                      // k == 0 for uncompressed lanes
        }
    } else if (AEC_ANY(kk > 8)) {
        // medium fields (16-bit uncompressed blocks among short codes): four of them per 64-bit peek
        constexpr uint32_t G4 = BS < 4 ? (uint32_t)BS : 4u;
        const uint32_t km = low_mask32(kk);
#pragma unroll
        for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += G4) {
            const bool gact = kk != 0 && g0 + G4 > off && g0 < nf + off;
            const uint32_t base = gact ? p + g0 * kk - off * kk : p;
            const uint64_t F = peek64(src, base);
#pragma unroll
            for (uint32_t j = 0; j < G4; j++) {
                const uint32_t i = g0 + j;
                const bool act = kk != 0 && i >= off && i - off < nf;
                const uint32_t f = (uint32_t)(F >> ((64u - (j + 1u) * kk) & 63u)) & km;
                d[i] = (u[i] << k) + (act ? f : 0u);
            }
        }
    } else {
        // narrow fields: eight of them per 64-bit peek
        const uint32_t km = low_mask32(kk);
#pragma unroll
        for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += GRP) {
            const bool gact = kk != 0 && g0 + GRP > off && g0 < nf + off;
            const uint32_t base = gact ? p + g0 * kk - off * kk : p;
            const uint64_t F = peek64(src, base);
#pragma unroll
            for (uint32_t j = 0; j < GRP; j++) {
                const uint32_t i = g0 + j;
                const bool act = kk != 0 && i >= off && i - off < nf;
                const uint32_t f = (uint32_t)(F >> ((64u - (j + 1u) * kk) & 63u)) & km;
                d[i] = (u[i] << k) + (act ? f : 0u);
            }
        }
    }
    p += nf * kk;
    if (ref && !unc) d[0] = refv;

    // ---- 4. second extension / zero run (rare, divergent) ------------------------------------
    if (AEC_ANY(se)) {
        if (se) {
            uint32_t i = ref;
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)BS / 2; j++) {
                const uint32_t m = ref ? u[1 + j] : u[j];   // j <= BS/2 - 1, so 1 + j < BS
                uint32_t s = 0, second = 0;
                if (!se_lookup(m, s, second)) corrupt = true;
                if ((i & 1u) == 0) d[i++] = s - second;
                d[i++] = second;
            }
            if (ref) d[0] = refv;
        }
    }
    if (AEC_ANY(zero)) {
        if (zero) {
            uint32_t nz = (ref ? u[1] : u[0]) + 1;
            if (nz == 5) {
                const uint32_t left_rsi = c.rsi - blk_in_rsi;
                const uint32_t left_seg = 64 - (blk_in_rsi % 64);
                nz = left_rsi < left_seg ? left_rsi : left_seg;
            } else if (nz > 5) {
                nz--;
            }
            if (nz > c.rsi - blk_in_rsi) corrupt = true;  // decode.c:543-544
            nzero_blocks = nz;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)BS; i++) d[i] = 0;
            if (ref) d[0] = refv;
        }
    }
    if (!live) return DEC_OK;
    if (short_input || p > end_p || src.starved()) return DEC_NEED_INPUT;
    return corrupt ? DEC_DATA_ERROR : DEC_OK;
}

// Same contract for blocks that are NOT the first of an RSI in any lane (ref == 0 everywhere: every
// block iteration but the first).  Without the per-lane reference slot the per-code and per-sample
// predication of decode_block_any disappears:
//   * unary phase: lanes whose option has no codes in a group of eight (uncompressed, zero blocks,
//     the upper half of a second-extension block) run the same instructions on an all-ones window,
//     which decodes as eight codes of value 0 and cannot trip the long-code test; their bit count
//     is simply not added.  The one code of a zero-block CDS is read in the rare zero branch.
//   * field phase: the field mask is 0 for k == 0, so no lane needs an "active" test.
template <int BS, class Src>
AEC_HD uint32_t decode_block_noref(Src &src, uint32_t &p, uint32_t end_p, uint32_t *d, const Cfg &c,
                                   uint32_t blk_in_rsi, bool live, uint32_t &nzero_blocks)
{
    static_assert(BS >= 2 && BS % 2 == 0, "templated block sizes only");
    const uint32_t idmax = (1u << c.id_len) - 1u;
    bool short_input = false, corrupt = false;
    nzero_blocks = 0;

    // ---- 1. header, 2. unary phase ------------------------------------------------------------
    // ONE 64-bit peek serves the header and the first group of up to 16 codes: at the k the reference
    // picks the unary region of a block is at most 3 bits per sample (see emit_small), so with blocks
    // of 8 or 16 samples the whole region lies inside it -- two LDS round trips (header, second group)
    // fewer on the serial path of a lane.  Zeros are shifted in from the right, so a code that runs out
    // of the window reads as 32+ zeros and takes the code-by-code path like any long code.
    uint32_t u[BS];
    // (larger blocks: groups of 8 -- 16 codes of a high-entropy block overrun the window too often)
    constexpr uint32_t GRP = BS <= AEC_DEC_GRP ? (uint32_t)BS : 8u;
    // second extension: BS/2 codes; where they end inside group 0, the slots behind them pick up bits
    // of the next coded data set, neither counted nor checked
    constexpr uint32_t SEH = ((uint32_t)BS / 2u) % GRP;
    const uint64_t U0 = peek64(src, p);
    const uint32_t h = (uint32_t)(U0 >> 32);
    const uint32_t id = h >> (32 - c.id_len);
    const bool lowent = live && id == 0;
    const bool unc = live && id == idmax;
    const bool split = live && !lowent && !unc;
    const uint32_t sel = (h >> (31 - c.id_len)) & 1u;
    const bool se = lowent && sel;
    const bool zero = lowent && !sel;
    const uint32_t k = split ? id - 1u : 0u;
    const uint32_t hdr = live ? c.id_len + (lowent ? 1u : 0u) : 0u;
    p += hdr;
#pragma unroll
    for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += GRP) {
        const bool part = SEH != 0 && g0 == 0 && se;
        const bool gact = split || (se && g0 < (uint32_t)BS / 2);
        uint64_t U = g0 == 0 ? (U0 << hdr) : peek64(src, p);
        if (!gact) U = ~0ull;
        // Per code: or, clz, add, or, 64-bit shift, add.  The or with 1 makes the clz of an empty upper
        // half read 31, and a code length of 32 (31 zeros -- or more) sets bit 5 of the or of all code
        // lengths: no min and no max on the way (the plain add / and / or / xor / right shift instructions
        // issue at twice the rate of everything else, profiles/r02/valu_issue_rate.txt), and a code of
        // exactly 31 zeros takes the code-by-code path along with the longer ones.
        uint32_t used = 0, lens = 0, used_h = 0, lens_h = 0;
#pragma unroll
        for (uint32_t j = 0; j < GRP; j++) {
            const uint32_t n1 = (uint32_t)__builtin_clz((uint32_t)(U >> 32) | 1u) + 1u;
            lens |= n1;
            U <<= (n1 & 63u);
            used += n1;
            u[g0 + j] = n1 - 1u;
            if (SEH != 0 && g0 == 0 && j == SEH - 1u) { used_h = used; lens_h = lens; }
        }
        if (part) { used = used_h; lens = lens_h; }
        const bool bad = (lens & 32u) != 0;              // a code of 31+ zeros: redo code by code
        if (AEC_ANY(bad)) {
            if (bad) {
                uint32_t q = p;
#pragma unroll
                for (uint32_t j = 0; j < GRP; j++)
                    if (!part || j < SEH) u[g0 + j] = unary_slow(src, q, end_p, short_input);
                used = q - p;
            }
        }
        p += gact ? used : 0u;
    }

    // ---- 3. field phase ---------------------------------------------------------------------
    const uint32_t kk = split ? k : (unc ? c.bps : 0u);
    if (AEC_ANY(kk > 16)) {
        // wide fields (uncompressed blocks, large k): one 32-bit peek per sample
        const uint32_t fsh = (32u - kk) & 31u;
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)BS; i++) {
            const uint32_t v = peek32(src, p + i * kk);
            const uint32_t f = kk != 0 ? v >> fsh : 0u;
            d[i] = (u[i] << k) + f;                      // k == 0 (and u == 0) for uncompressed lanes
        }
    } else if (AEC_ANY(kk > 8)) {
        // medium fields (16-bit uncompressed blocks among short codes): four of them per 64-bit peek
        constexpr uint32_t G4 = BS < 4 ? (uint32_t)BS : 4u;
        const uint32_t km = low_mask32(kk);
#pragma unroll
        for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += G4) {
            const uint64_t F = peek64(src, p + g0 * kk);
#pragma unroll
            for (uint32_t j = 0; j < G4; j++) {
                const uint32_t f = (uint32_t)(F >> ((64u - (j + 1u) * kk) & 63u)) & km;
                d[g0 + j] = (u[g0 + j] << k) + f;
            }
        }
    } else {
        // narrow fields: eight of them per 64-bit peek, each taken from the top of the window (a right
        // shift of the upper half) before the window moves up by one field
        constexpr uint32_t FG = BS < 8 ? (uint32_t)BS : 8u;
        const uint32_t fsh = (32u - kk) & 31u, fm = kk ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (uint32_t g0 = 0; g0 < (uint32_t)BS; g0 += FG) {
            uint64_t F = peek64(src, p + g0 * kk);
#pragma unroll
            for (uint32_t j = 0; j < FG; j++) {
                const uint32_t f = ((uint32_t)(F >> 32) >> fsh) & fm;
                F <<= kk;
                d[g0 + j] = (u[g0 + j] << k) + f;
            }
        }
    }
    p += (uint32_t)BS * kk;

    // ---- 4. second extension / zero run (rare, divergent) ------------------------------------
    if (AEC_ANY(se)) {
        if (se) {
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)BS / 2; j++) {
                uint32_t s = 0, second = 0;
                if (!se_lookup(u[j], s, second)) corrupt = true;
                d[2 * j] = s - second;
                d[2 * j + 1] = second;
            }
        }
    }
    if (AEC_ANY(zero)) {
        if (zero) {
            uint32_t nz = unary_slow(src, p, end_p, short_input) + 1;
            if (nz == 5) {
                const uint32_t left_rsi = c.rsi - blk_in_rsi;
                const uint32_t left_seg = 64 - (blk_in_rsi % 64);
                nz = left_rsi < left_seg ? left_rsi : left_seg;
            } else if (nz > 5) {
                nz--;
            }
            if (nz > c.rsi - blk_in_rsi) corrupt = true;  // decode.c:543-544
            nzero_blocks = nz;
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)BS; i++) d[i] = 0;
        }
    }
    if (!live) return DEC_OK;
    if (short_input || p > end_p || src.starved()) return DEC_NEED_INPUT;
    return corrupt ? DEC_DATA_ERROR : DEC_OK;
}

// One block per lane.  `ref` != 0 marks the first block of an RSI (it carries the reference
// sample); a wave without such a lane takes the leaner decode_block_noref.
template <int BS, class Src>
AEC_HD uint32_t decode_block(Src &src, uint32_t &p, uint32_t end_p, uint32_t *d, const Cfg &c,
                             uint32_t ref, uint32_t blk_in_rsi, bool live, uint32_t &nzero_blocks)
{
    if (AEC_ANY(ref != 0)) return decode_block_any<BS>(src, p, end_p, d, c, ref, blk_in_rsi, live, nzero_blocks);
    return decode_block_noref<BS>(src, p, end_p, d, c, blk_in_rsi, live, nzero_blocks);
}

// Skips one CDS without materialising samples (RSI index pass).  Returns blocks covered in
// `nblocks` (1, or the zero-run length).
template <class Reader>
AEC_HD uint32_t skip_cds(Reader &r, const Cfg &c, uint32_t ref, uint32_t blk_in_rsi,
                         uint32_t &nblocks)
{
    nblocks = 1;
    const uint32_t id = r.get(c.id_len);
    if (id == 0) {
        const uint32_t sel = r.get(1);
        if (ref) r.skip(c.bps);
        if (sel) {
            // every SE code must be a valid table entry; validate while skipping
            for (uint32_t j = 0; j < c.bs / 2; j++) {
                uint32_t m;
                if (!r.unary(m)) return DEC_NEED_INPUT;
                if (m > 90) return r.overrun() ? DEC_NEED_INPUT : DEC_DATA_ERROR;
            }
        } else {
            uint32_t fs;
            if (!r.unary(fs)) return DEC_NEED_INPUT;
            uint32_t nz = fs + 1;
            if (nz == 5) {
                const uint32_t left_rsi = c.rsi - blk_in_rsi;
                const uint32_t left_seg = 64 - (blk_in_rsi % 64);
                nz = left_rsi < left_seg ? left_rsi : left_seg;
            } else if (nz > 5) {
                nz--;
            }
            if (r.overrun()) return DEC_NEED_INPUT;
            if (nz > c.rsi - blk_in_rsi) return DEC_DATA_ERROR;
            nblocks = nz;
        }
    } else if (id == (1u << c.id_len) - 1u) {
        r.skip((uint64_t)c.bs * c.bps);
    } else {
        const uint32_t k = id - 1;
        if (ref) r.skip(c.bps);
        if (!r.skip_unary(c.bs - ref)) return DEC_NEED_INPUT;
        r.skip((uint64_t)(c.bs - ref) * k);
    }
    return r.overrun() ? DEC_NEED_INPUT : DEC_OK;
}

}  // namespace aec
