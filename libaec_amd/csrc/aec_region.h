// aec_region.h -- per-lane arithmetic of the REGION index (aec_region.hip; DESIGN.md section 2, scheme "regions").
//
// A bare stream of coded data sets has no entry points (reference src/decode.c:402-421), but a lane that stands on an
// RSI start walks the stream from there at a few hundred instructions per coded data set, and a chip full of such lanes
// passes over a gigabyte in a millisecond or two.  What a lane in the middle of the stream lacks is the place to stand
// on.  The stream goes in regions; every region but the first GUESSES its entry -- the first RSI start behind its
// first bit that it can recognise -- from what real data looks like (below); every region is walked from its entry with
// the RSI's own bookkeeping (reference sample every `rsi` blocks, zero runs by the block count: decode.c:518-558) up to
// the entry of the region behind it; a region whose entry is not the place where the walk in front of it arrived is
// walked again from there.  Region 0 begins on the caller's exact state, so by induction every entry that survives is
// the reference's own walk: nothing that is delivered rests on a guess, a wrong guess costs one more walk of a region.
//
// The guess (RgGuess).  With the preprocessor the first coded data set of an RSI carries the raw reference sample
// (decode.c:407-411 `ref = pp`, :462-470), and neighbouring blocks of real data are coded with neighbouring options
// (the option is the magnitude of the residuals: encode.c:313-410), while a parse from a wrong bit reads id_len
// random bits.  Every coded data set of a chain gets a SCORE from the distance of its option to the one in front
// (rg_score: about the log-likelihood ratio "true chain" against "random bits").  Then
//   1. ANCHOR: chains without reference samples from consecutive bits; one whose score reaches kRgAnchorScore before
//      it falls to kRgAnchorFail stands on the true chain (a chain that started beside it has fallen into it on
//      the way, or fails);
//   2. WALK on without reference samples.  The coded data set that DOES hold a reference sample throws the parse off
//      the true chain and the options turn random: a leaky sum of the scores falls below kRgSuspect within a few
//      coded data sets;
//   3. TEST the last kRgHist boundaries as RSI starts, the latest first: the coded data set parsed WITH a reference
//      sample, then kRgTest more, scored the same way.  The first candidate that reaches kRgTestPass is the guess.  None:
//      the data made a jump (on with the walk), or -- three times in a row -- the chain is lost (a new anchor).
// An RSI start that the plain parse survives (an uncompressed first block holds its reference sample as sample 0,
// decode.c:659-677; a lucky landing) is simply passed: the next one is rsi blocks on.
// The guess is a state machine that takes ONE coded data set per step whatever it is doing: the lanes of a wavefront
// are in different states, and this way they share the parse.
//
// Everything here is __host__ __device__: tests/emul/region_emul.cpp runs the same functions on the CPU against the
// RSI starts the oracle's encoder reports.
#pragma once

#include "aec_trunk.h"

namespace aec {

constexpr int32_t kRgAnchorScore = 16, kRgAnchorFail = -4;
constexpr uint32_t kRgHist = 12;           // boundaries the walk remembers (RgGuess::hp0 .. hp11)
constexpr uint32_t kRgCand = 3;            // ... of which the three in front of the first option that did not fit are tried as the RSI start
constexpr uint32_t kRgDefer = 3;           // coded data sets the walk goes on before it tests (a jump of the data: the sum recovers)
constexpr uint32_t kRgTest = 16;           // coded data sets behind a candidate's first
constexpr int32_t kRgTestPass = 18, kRgTestFail = -8;
constexpr int32_t kRgMargin = 8;           // ... and by how much the chain WITH a reference sample must beat the one without
constexpr uint32_t kRgVerifyMaxRsi = 512;  // RSIs of up to this many blocks: the candidate's RSI is walked and its END tested too
constexpr int32_t kRgTestPass2 = 12, kRgMargin2 = 0;
constexpr int32_t kRgLeakStart = 32, kRgSuspect = -8, kRgHealthy = 12;    // leaky sum, in quarters

// score of an option `id` behind the option `prev`: near log2 of P(distance | true chain) / P(distance | random bits)
AEC_HD int32_t rg_score(uint32_t id, uint32_t prev, uint32_t id_len)
{
    const uint32_t d = id > prev ? id - prev : prev - id;
    if (d <= 1u) return 2;
    if (id_len >= 5u) return d == 2u ? 1 : (d == 3u ? -2 : -6);
    return d == 2u ? -2 : -6;
}

// ---- the lane's view of the stream ----------------------------------------------------------------------------------
// 384 bits of the stream in registers from a bit position that is a multiple of 64 on: the coded data set at hand is
// parsed out of the first 256, the last 128 are the load in flight.  A lane that walks a chain reads every byte of its
// stretch once or twice, sixteen at a time, and the load for the steps to come is on its way while this one is parsed
// (tr_cds loads 32 bytes per coded data set and waits for them).  A coded data set whose unary part ends inside 128 bits
// comes out of the registers (~110 instructions); anything else is tr_cds' from memory (exact either way: rg_cds
// returns what tr_cds returns, tests/emul/region_emul.cpp compares them bit by bit).
struct RgWin {
    uint64_t w0, w1, w2, w3;   // (named, not an array: indexing by a lane value would put it in scratch)
    uint64_t w4, w5;           // bits [wb + 256, wb + 384): in flight until the window moves on
    uint64_t wb;               // bit position of w0's first bit, multiple of 64
};

// 64 stream bits from the 64-bit group g (bit position 64 g) on; groups beyond the buffer repeat its last word
AEC_HD uint64_t rg_group(const TrStream &s, uint64_t g)
{
    const uint64_t w = g << 1;
    return ((uint64_t)tr_word(s, w) << 32) | tr_word(s, w + 1u);
}

// groups g and g + 1
AEC_HD void rg_group2(const TrStream &s, uint64_t g, uint64_t &a, uint64_t &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if ((g << 1) + 4u <= s.nwords) {
        struct __attribute__((packed, aligned(4))) Q4 {
            uint32_t a, b, c, d;
        };
        const Q4 v = *reinterpret_cast<const Q4 *>(s.words + (g << 1));
        a = ((uint64_t)bswap32(v.a) << 32) | bswap32(v.b);
        b = ((uint64_t)bswap32(v.c) << 32) | bswap32(v.d);
        return;
    }
#endif
    a = rg_group(s, g);
    b = rg_group(s, g + 1u);
}

AEC_HD void rg_seat(const TrStream &s, RgWin &W, uint64_t q)
{
    const uint64_t g = q >> 6;
    W.wb = g << 6;
    rg_group2(s, g, W.w0, W.w1);
    rg_group2(s, g + 2u, W.w2, W.w3);
    rg_group2(s, g + 4u, W.w4, W.w5);
}

// the window so that wb <= q < wb + 64
AEC_HD void rg_window(const TrStream &s, RgWin &W, uint64_t q)
{
    const uint64_t d = q - W.wb;                 // (q < wb: a huge number)
    if (d < 64u) return;
    if (d < 192u) {                              // one or two groups on: the load in flight lands, the next one leaves
        const bool two = d >= 128u;
        W.w0 = two ? W.w2 : W.w1;
        W.w1 = two ? W.w3 : W.w2;
        W.w2 = two ? W.w4 : W.w3;
        W.w3 = two ? W.w5 : W.w4;
        W.wb += two ? 128u : 64u;
        rg_group2(s, (W.wb >> 6) + 4u, W.w4, W.w5);
        return;
    }
    rg_seat(s, W, q);
}

// 64 bits at bit offset o < 192 of the 256 bits a : b : c : d
AEC_HD uint64_t rg_peek4(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint32_t o)
{
    const uint32_t g = o >> 6, sh = o & 63u;
    const uint64_t hi = g == 0u ? a : (g == 1u ? b : c), lo = g == 0u ? b : (g == 1u ? c : d);
    return sh ? (hi << sh) | (lo >> (64u - sh)) : hi;
}

// tr_cds(s, c, q, ref, nz) and the option id of the coded data set at q; moves the window to q
AEC_HD uint32_t rg_cds(const TrStream &s, const Cfg &c, RgWin &W, uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz)
{
    nz = 0;
    id = 0;
    if (q + c.id_len >= s.end_bit) return 0;
    rg_window(s, W, q);
    const uint32_t o = (uint32_t)(q - W.wb);      // < 64
    const uint64_t H = rg_peek4(W.w0, W.w1, W.w2, W.w3, o);
    id = (uint32_t)(H >> (64u - c.id_len));
    if (id == (1u << c.id_len) - 1u) {
        const uint32_t len = c.id_len + c.bs * c.bps;
        return q + len <= s.end_bit ? len : 0u;
    }
    const bool low = id == 0u;
    const uint32_t selbit = (uint32_t)(H >> (63u - c.id_len)) & 1u;
    const uint32_t hdr = c.id_len + (low ? 1u : 0u) + ref * c.bps;          // <= 38
    const uint32_t need = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
    const uint32_t add = low ? 0u : need * (id - 1u);
    const uint32_t o2 = o + hdr;                                            // < 102: two pieces end below bit 230
    const uint64_t U0 = rg_peek4(W.w0, W.w1, W.w2, W.w3, o2);
    const uint32_t p0 = tr_popc64(U0);
    uint32_t used;
    if (p0 >= need) {
        used = hdr + spec_select64(U0, need) + 1u;
    } else {
        const uint64_t U1 = rg_peek4(W.w0, W.w1, W.w2, W.w3, o2 + 64u);
        const uint32_t p1 = tr_popc64(U1);
        if (p0 + p1 < need) return tr_cds(s, c, q, ref, nz);                // (rare: from memory)
        used = hdr + 64u + spec_select64(U1, need - p0) + 1u;
    }
    if (low && !selbit) nz = used - hdr;
    const uint64_t len = (uint64_t)used + add;
    return q + len <= s.end_bit ? (uint32_t)len : 0u;
}

// The parser interface of the functions below: uint32_t cds(uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz),
// length in bits, 0 = no coded data set ends inside the stream from q.
struct RgLaneParser {
    const TrStream &s;
    const Cfg &c;
    RgWin W;
    AEC_HD void seat(uint64_t q) { rg_seat(s, W, q); }
    AEC_HD uint32_t cds(uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz) { return rg_cds(s, c, W, q, ref, id, nz); }
};

// ---- the lane's ring ------------------------------------------------------------------------------------------------
// The register window above asks memory for 16 bytes whenever a lane crosses a 64-bit group, and waits for them a step
// later.  A wavefront of 64 lanes then waits at every step for whichever of its lanes touched a new line of HBM: 2.7 us
// per coded data set, measured, whatever the lane itself needed.  Here a lane keeps kRgRingWords words of its stretch
// of the stream in a ring (LDS on the device: word i of the stream at ring[(i mod kRgRingWords) * stride], a column per
// lane, conflict-free), parses out of it (seven words: header, 128 bits of unary part), and the WAVEFRONT tops the rings
// up together every `period` steps: up to eight 16-byte loads per lane leave, and land at the next top-up, `period` steps
// -- microseconds -- later: what memory takes is over by then for every lane.  The ring keeps kRgRingBack words behind
// the parse (the guess goes back a few coded data sets for its tests); a jump out of the ring, or a lane that consumed
// faster than the top-ups bring, fills up at once (rare, and the whole wavefront waits).
constexpr uint32_t kRgRingWords = 64, kRgRingBack = 16, kRgRingChunks = 8;

// offset (from the most significant bit, 0-based) of the n-th 1-bit of v, 1 <= n <= popcount(v)
AEC_HD uint32_t rg_select32(uint32_t v, uint32_t n)
{
    uint32_t pos = 0, cnt;
    cnt = spec_popc(v >> 16);
    if (n > cnt) { n -= cnt; pos = 16; v &= 0xFFFFu; } else { v >>= 16; }
    cnt = spec_popc(v >> 8);
    if (n > cnt) { n -= cnt; pos += 8; v &= 0xFFu; } else { v >>= 8; }
    cnt = spec_popc(v >> 4);
    if (n > cnt) { n -= cnt; pos += 4; v &= 0xFu; } else { v >>= 4; }
    cnt = spec_popc(v >> 2);
    if (n > cnt) { n -= cnt; pos += 2; v &= 0x3u; } else { v >>= 2; }
    cnt = v >> 1;
    if (n > cnt) pos += 1;
    return pos;
}

struct RgChunk {
    uint32_t a, b, c, d;      // four stream words as they lie in memory
};

struct RgRingParser {
    const TrStream &s;
    const Cfg &c;
    uint32_t *ring;
    uint32_t stride;
    uint32_t period;           // steps between top-ups
    uint32_t tick;
    uint64_t lo, hi;           // stream words [lo, hi) are in the ring; multiples of 4, hi - lo <= kRgRingWords
    uint32_t pv;               // chunks in flight: words [hi, hi + 4 pv)
    bool aligned16;            // the stream's buffer begins on a 16-byte boundary (chunks are one load each)
    RgChunk p0, p1, p2, p3, p4, p5, p6, p7;      // (named: see RgGuess::hp0)

    AEC_HD RgChunk load_chunk(uint64_t w) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (w + 4u <= s.nwords && aligned16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(s.words + w);
            return RgChunk{v.x, v.y, v.z, v.w};
        }
        if (w + 4u <= s.nwords) return RgChunk{s.words[w], s.words[w + 1u], s.words[w + 2u], s.words[w + 3u]};
#endif
        // (words beyond the buffer repeat its last one, as tr_word has it)
        return RgChunk{bswap32(tr_word(s, w)), bswap32(tr_word(s, w + 1u)), bswap32(tr_word(s, w + 2u)), bswap32(tr_word(s, w + 3u))};
    }
    AEC_HD void put(uint64_t w, const RgChunk &v)
    {
        uint32_t *q = ring + (size_t)((uint32_t)w & (kRgRingWords - 1u)) * stride;      // (w is a multiple of 4: no wrap inside)
        q[0] = bswap32(v.a);
        q[stride] = bswap32(v.b);
        q[2u * stride] = bswap32(v.c);
        q[3u * stride] = bswap32(v.d);
    }
    AEC_HD uint32_t word(uint64_t w) const { return ring[(size_t)((uint32_t)w & (kRgRingWords - 1u)) * stride]; }
    AEC_HD void land()
    {
        if (pv > 0u) put(hi, p0);
        if (pv > 1u) put(hi + 4u, p1);
        if (pv > 2u) put(hi + 8u, p2);
        if (pv > 3u) put(hi + 12u, p3);
        if (pv > 4u) put(hi + 16u, p4);
        if (pv > 5u) put(hi + 20u, p5);
        if (pv > 6u) put(hi + 24u, p6);
        if (pv > 7u) put(hi + 28u, p7);
        hi += 4u * pv;
        pv = 0;
    }
    // words behind wq that the parse will not ask for again make room
    AEC_HD void trim(uint64_t wq)
    {
        const uint64_t keep = (wq > kRgRingBack ? wq - kRgRingBack : 0u) & ~3ull;
        if (keep > lo) lo = keep < hi ? keep : hi;
    }
    AEC_HD void issue()
    {
        const uint32_t room = (kRgRingWords - (uint32_t)(hi - lo)) >> 2;
        // (nothing behind the end of the buffer is worth a load: the parse there reads the repeated last word)
        const uint32_t n = hi >= s.nwords + 8u ? 0u : (room < kRgRingChunks ? room : kRgRingChunks);
        pv = n;
        if (n > 0u) p0 = load_chunk(hi);
        if (n > 1u) p1 = load_chunk(hi + 4u);
        if (n > 2u) p2 = load_chunk(hi + 8u);
        if (n > 3u) p3 = load_chunk(hi + 12u);
        if (n > 4u) p4 = load_chunk(hi + 16u);
        if (n > 5u) p5 = load_chunk(hi + 20u);
        if (n > 6u) p6 = load_chunk(hi + 24u);
        if (n > 7u) p7 = load_chunk(hi + 28u);
    }
    AEC_HD void seat(uint64_t q)
    {
        lo = hi = (q >> 5) & ~3ull;                      // (the next parse finds its words missing and fills up)
        pv = 0;
        tick = 0;
    }
    AEC_HD void init(uint32_t *ring_, uint32_t stride_, uint32_t period_)
    {
        ring = ring_;
        stride = stride_;
        period = period_ ? period_ : 1u;
        tick = 0;
        lo = hi = 0;
        pv = 0;
        aligned16 = (reinterpret_cast<uintptr_t>(s.words) & 15u) == 0u;
    }

    // rg_cds' results (= tr_cds' and the option id) out of the ring.  32-bit arithmetic throughout: this is the inner
    // loop of every pass of the scheme (~100 instructions for a coded data set whose unary part ends inside 128 bits).
    AEC_HD uint32_t cds(uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz)
    {
        nz = 0;
        id = 0;
        if (q + c.id_len >= s.end_bit) return 0;
        const uint64_t wq = q >> 5;
        // the wavefront's top-up (its lanes step together) -- or this lane's own: it jumped out of its ring (and of
        // what is on its way), or consumed faster than the top-ups bring; then the loop below comes round again and waits
        const uint32_t dl = (uint32_t)wq - (uint32_t)lo, span = (uint32_t)hi - (uint32_t)lo;   // (behind lo: huge)
        const bool outside = dl > span + 4u * pv;
        if (++tick >= period || outside || dl + 7u > span) {
            tick = 0;
            if (outside) {
                pv = 0;
                lo = hi = wq & ~3ull;
            }
            for (;;) {
                land();
                trim(wq);
                issue();
                if (wq + 7u <= hi) break;
            }
        }
        const uint32_t w0 = word(wq), w1 = word(wq + 1u), w2 = word(wq + 2u), w3 = word(wq + 3u), w4 = word(wq + 4u),
                       w5 = word(wq + 5u), w6 = word(wq + 6u);
        const uint64_t left64 = s.end_bit - q;
        const uint32_t left = left64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left64;
        const uint32_t sh = (uint32_t)q & 31u;
        const uint32_t h = spec_shl_hi(w0, w1, sh);
        id = h >> (32u - c.id_len);
        if (id == (1u << c.id_len) - 1u) {
            const uint32_t len = c.id_len + c.bs * c.bps;
            return len <= left ? len : 0u;
        }
        const bool low = id == 0u;
        const uint32_t selbit = (h >> (31u - c.id_len)) & 1u;
        const uint32_t hdr = c.id_len + (low ? 1u : 0u) + ref * c.bps;          // <= 38
        const uint32_t need = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
        const uint32_t add = low ? 0u : need * (id - 1u);
        const uint32_t o2 = sh + hdr, i = o2 >> 5, t2 = o2 & 31u;               // o2 < 70: i <= 2
        const uint32_t a0 = i == 0u ? w0 : (i == 1u ? w1 : w2), a1 = i == 0u ? w1 : (i == 1u ? w2 : w3),
                       a2 = i == 0u ? w2 : (i == 1u ? w3 : w4), a3 = i == 0u ? w3 : (i == 1u ? w4 : w5),
                       a4 = i == 0u ? w4 : (i == 1u ? w5 : w6);
        // 128 bits of unary part in four pieces; their running counts of 1-bits say which holds the need-th
        const uint32_t u0 = spec_shl_hi(a0, a1, t2), u1 = spec_shl_hi(a1, a2, t2), u2 = spec_shl_hi(a2, a3, t2),
                       u3 = spec_shl_hi(a3, a4, t2);
        const uint32_t c0 = spec_popc(u0), c1 = c0 + spec_popc(u1), c2 = c1 + spec_popc(u2), c3 = c2 + spec_popc(u3);
        if (c3 < need) return tr_cds(s, c, q, ref, nz);                         // (rare: from memory)
        const uint32_t k = (need > c0 ? 1u : 0u) + (need > c1 ? 1u : 0u) + (need > c2 ? 1u : 0u);
        const uint32_t u = k == 0u ? u0 : (k == 1u ? u1 : (k == 2u ? u2 : u3));
        const uint32_t before = k == 0u ? 0u : (k == 1u ? c0 : (k == 2u ? c1 : c2));
        const uint32_t used = hdr + 32u * k + rg_select32(u, need - before) + 1u;
        if (low && !selbit) nz = used - hdr;
        const uint32_t len = used + add;
        return len <= left ? len : 0u;
    }
};

// steps between the top-ups of the rings: so that eight chunks (128 bytes) are twice what a lane consumes in between
AEC_HD uint32_t rg_ring_period(uint64_t avg_cds_bits)
{
    const uint64_t p = avg_cds_bits ? (128u * 8u) / (2u * avg_cds_bits) : 8u;
    return p < 1u ? 1u : (p > 8u ? 8u : (uint32_t)p);
}

struct RgMemParser {                       // straight from memory (what the lane parser is checked against)
    const TrStream &s;
    const Cfg &c;
    AEC_HD void seat(uint64_t) {}
    AEC_HD uint32_t cds(uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz) const
    {
        nz = 0;
        id = 0;
        if (q + c.id_len >= s.end_bit) return 0;
        id = (uint32_t)(tr_peek64(s, q) >> (64u - c.id_len));
        return tr_cds(s, c, q, ref, nz);
    }
};

struct RgGuess {
    enum : uint32_t { ANCHOR = 0, WALK = 1, TEST = 2, VERIFY = 3, FOUND = 4, NONE = 5 };
    uint32_t mode;
    uint64_t q;                // where the next parse begins
    uint32_t ref;              // ... with a reference sample
    uint32_t prev;             // option in front of q; bit 8: that coded data set was a run of zero blocks
    int32_t S;                 // score of the chain at hand (anchor, test)
    uint32_t steps;
    uint64_t t;                // anchor: first bit of the chain at hand
    uint64_t base;             // positions of the history are relative to this
    // walk: the last boundaries (hp0 the latest), relative to base, ~0 = none; and `prev` in front of each
    // (named, not arrays: the compiler keeps a struct with arrays in scratch memory)
    uint32_t hp0, hp1, hp2, hp3, hp4, hp5, hp6, hp7, hp8, hp9, hp10, hp11;
    uint32_t ho0, ho1, ho2, ho3, ho4, ho5, ho6, ho7, ho8, ho9, ho10, ho11;
    int32_t L;                 // leaky sum of the walk's scores
    uint32_t fails;            // tests in a row that found nothing
    uint32_t defer;            // walk: coded data sets until the test (0: no suspicion)
    uint32_t age;              // walk: coded data sets since the first option of this stretch that did not fit, that one included (0: none)
    uint32_t jend;             // test: one past the last candidate
    uint32_t j;                // test: candidate at hand
    uint32_t sub;              // test: 0 = the chain with a reference sample, 1 = the one without from the same boundary
    int32_t Sg;                // test: score of the first of the two
    int32_t best;              // test: best score of a candidate so far, and which
    uint32_t best_j;
    uint64_t wq;               // test: where the walk goes on
    uint32_t wprev;
    uint32_t vb;               // verify: blocks of the candidate's RSI done
    uint32_t second;           // test: 1 = at the END of the candidate's RSI (the next RSI start, if the candidate is one)
    uint64_t cq;               // test: the boundary under test
    uint32_t cprev;
    uint64_t found;            // FOUND: the RSI start
    uint32_t parses;

    AEC_HD void init(uint64_t from)
    {
        mode = ANCHOR;
        q = t = base = from;
        ref = 0;
        prev = 0;
        S = 0;
        steps = 0;
        hist_clear();
        ho0 = ho1 = ho2 = ho3 = ho4 = ho5 = ho6 = ho7 = ho8 = ho9 = ho10 = ho11 = 0u;
        L = 0;
        fails = 0;
        defer = 0;
        age = 0;
        jend = 0;
        j = 0;
        sub = 0;
        Sg = 0;
        best = 0;
        best_j = 0;
        wq = 0;
        wprev = 0;
        vb = 0;
        second = 0;
        cq = 0;
        cprev = 0;
        found = 0;
        parses = 0;
    }
    AEC_HD bool busy() const { return mode < FOUND; }

    // score of the coded data set (id, nz) behind `prev`, and `prev` behind it
    AEC_HD static int32_t score(const Cfg &c, uint32_t len, uint32_t id, uint32_t nz, uint32_t &prev)
    {
        if (!len) return -1000;
        if (id == 0u && nz != 0u) {
            // a run of zero blocks fits behind any option (a constant stretch begins where it likes) -- but the encoder
            // writes ONE coded data set per run up to the end of the segment (encode.c:614-659): a run behind a run is
            // the rest-of-segment code, or garbage
            const int32_t sc = ((prev & 0x100u) && nz != 5u) ? -4 : 0;
            prev |= 0x100u;
            return sc;
        }
        const int32_t sc = rg_score(id, prev & 0xFFu, c.id_len);
        prev = id;
        return sc;
    }
    AEC_HD void new_anchor(uint64_t at, uint64_t end_bit, const Cfg &c)
    {
        mode = at + c.id_len < end_bit ? ANCHOR : NONE;
        t = q = at;
        ref = 0;
        S = 0;
        steps = 0;
    }
    AEC_HD void hist_clear() { hp0 = hp1 = hp2 = hp3 = hp4 = hp5 = hp6 = hp7 = hp8 = hp9 = hp10 = hp11 = ~0u; }
    AEC_HD void hist_push(uint32_t p, uint32_t o)
    {
        hp11 = hp10; hp10 = hp9; hp9 = hp8; hp8 = hp7; hp7 = hp6; hp6 = hp5; hp5 = hp4; hp4 = hp3; hp3 = hp2; hp2 = hp1; hp1 = hp0;
        ho11 = ho10; ho10 = ho9; ho9 = ho8; ho8 = ho7; ho7 = ho6; ho6 = ho5; ho5 = ho4; ho4 = ho3; ho3 = ho2; ho2 = ho1; ho1 = ho0;
        hp0 = p;
        ho0 = o;
    }
    AEC_HD static uint32_t pick(uint32_t k, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11)
    {
        uint32_t v = a0;
        v = k == 1u ? a1 : v;
        v = k == 2u ? a2 : v;
        v = k == 3u ? a3 : v;
        v = k == 4u ? a4 : v;
        v = k == 5u ? a5 : v;
        v = k == 6u ? a6 : v;
        v = k == 7u ? a7 : v;
        v = k == 8u ? a8 : v;
        v = k == 9u ? a9 : v;
        v = k == 10u ? a10 : v;
        v = k == 11u ? a11 : v;
        return v;
    }
    AEC_HD uint32_t hist_pos(uint32_t k) const { return pick(k, hp0, hp1, hp2, hp3, hp4, hp5, hp6, hp7, hp8, hp9, hp10, hp11); }
    AEC_HD uint32_t hist_opt(uint32_t k) const { return pick(k, ho0, ho1, ho2, ho3, ho4, ho5, ho6, ho7, ho8, ho9, ho10, ho11); }
    // the suspicion came to nothing: on with the walk, or -- the third time in a row -- the chain is lost
    AEC_HD void resume_walk(uint64_t end_bit, const Cfg &c)
    {
        fails++;
        if (fails >= 3u) {
            new_anchor(wq + 1u, end_bit, c);
            return;
        }
        mode = WALK;
        q = wq;
        ref = 0;
        prev = wprev;
        L = 0;
        defer = 0;
        age = 0;
    }
    AEC_HD void test_begin(uint64_t at, uint32_t opt_in_front)
    {
        mode = TEST;
        cq = q = at;
        cprev = prev = opt_in_front;
        ref = 1;
        sub = 0;
        S = 0;
        steps = 0;
    }
    AEC_HD void next_candidate(uint64_t end_bit, const Cfg &c)
    {
        j++;
        const uint32_t p = (j < kRgHist && j < jend) ? hist_pos(j) : ~0u;
        if (p != ~0u) {
            test_begin(base + p, hist_opt(j));
            return;
        }
        // all tried
        if (best < kRgTestPass) {
            resume_walk(end_bit, c);
            return;
        }
        found = base + hist_pos(best_j);
        if (c.rsi > kRgVerifyMaxRsi) {
            mode = FOUND;
            return;
        }
        // short RSIs: the candidate's whole RSI, and the same test where it ends
        mode = VERIFY;
        q = found;
        ref = 1;
        prev = hist_opt(best_j);
        vb = 0;
    }

    // one coded data set: (len, id, nz) = the parse at q with `ref`
    AEC_HD void step(const Cfg &c, uint64_t end_bit, uint32_t len, uint32_t id, uint32_t nz)
    {
        parses++;
        if (mode == ANCHOR) {
            uint32_t pv = steps ? prev : id;             // (the first has nothing in front of it)
            const int32_t sc = score(c, len, id, nz, pv);
            S += sc;
            if (!len || S <= kRgAnchorFail) {
                new_anchor(t + 1u, end_bit, c);
                return;
            }
            prev = pv;
            steps++;
            q += len;
            if (S >= kRgAnchorScore) {
                mode = WALK;
                L = kRgLeakStart;
                fails = 0;
                base = q;
                hist_clear();
                defer = 0;
                age = 0;
            }
            return;
        }
        if (mode == WALK) {
            if (!len) {
                new_anchor(q + 1u, end_bit, c);
                return;
            }
            const uint32_t was = prev;
            const int32_t sc = score(c, len, id, nz, prev);
            L = L - (L >> 2) + 4 * sc;
            if (L >= kRgHealthy) fails = 0;
            age = age ? age + 1u : (sc < 0 ? 1u : 0u);
            if (L >= kRgHealthy && sc > 0) age = 0u;
            hist_push((uint32_t)(q - base), was);
            q += len;
            if (defer) {
                if (--defer == 0u && L < kRgHealthy) {
                    // the RSI start is the boundary in front of the first coded data set that did not fit (hp[age]), or --
                    // the first garbage options fitted by chance -- one of the two in front of that
                    wq = q;
                    wprev = prev;
                    const uint32_t first = age ? age : 1u;
                    j = first - 1u;
                    jend = first + kRgCand;
                    best = -1000;
                    second = 0;
                    next_candidate(end_bit, c);
                }
            } else if (L < kRgSuspect) {
                defer = kRgDefer;
            }
            return;
        }
        if (mode == VERIFY) {
            const uint32_t nb = len ? tr_blocks(c, nz, vb) : 0u;
            if (!nb) {
                resume_walk(end_bit, c);
                return;
            }
            (void)score(c, len, id, nz, prev);
            vb += nb;
            q += len;
            ref = 0;
            if (vb >= c.rsi) {
                second = 1;
                test_begin(q, prev);
            }
            return;
        }
        // TEST
        S += score(c, len, id, nz, prev);
        const int32_t pass = second ? kRgTestPass2 : kRgTestPass;
        if (sub == 0u) {
            if (S <= kRgTestFail || (steps == kRgTest && (S < pass || (!second && S <= best)))) {
                if (second)
                    resume_walk(end_bit, c);
                else
                    next_candidate(end_bit, c);  // (ties between candidates: the later boundary)
                return;
            }
            if (steps == kRgTest) {        // good so far: the same boundary WITHOUT a reference sample
                Sg = S;
                sub = 1;
                q = cq;
                ref = 0;
                prev = cprev;
                S = 0;
                steps = 0;
                return;
            }
        } else if (steps == kRgTest || S <= kRgTestFail) {
            const bool ok = Sg - S >= (second ? kRgMargin2 : kRgMargin);
            if (second) {
                if (ok)
                    mode = FOUND;
                else
                    resume_walk(end_bit, c);
                return;
            }
            if (ok) {
                best = Sg;
                best_j = j;
            }
            next_candidate(end_bit, c);
            return;
        }
        steps++;
        q += len;
        ref = 0;
    }
};

struct RgGuessStats {                      // (emulator / tuning builds)
    uint64_t parses, found;
};

// The first RSI start the guess from bit `from` recognises within `budget` coded data sets
// (in front of `limit`: the walk of the guess does not go beyond it).
template <class P>
AEC_HD bool rg_guess(P &ps, const Cfg &c, uint64_t from, uint64_t limit, uint64_t end_bit, uint32_t budget,
                     uint64_t &rsi_start, RgGuessStats *stats = nullptr)
{
    RgGuess g;
    g.init(from);
    if (from + c.id_len >= end_bit) g.mode = RgGuess::NONE;
    ps.seat(from);
    while (g.busy() && g.parses < budget) {
        if (g.mode <= RgGuess::WALK && g.q >= limit) break;
        uint32_t id, nz;
        const uint32_t len = ps.cds(g.q, g.ref, id, nz);
        g.step(c, end_bit, len, id, nz);
    }
    if (stats) stats->parses += g.parses;
    if (g.mode != RgGuess::FOUND) return false;
    rsi_start = g.found;
    return true;
}

// ---- the walk with the RSI's bookkeeping ----------------------------------------------------------------------------
struct RgState {
    uint64_t pos;
    uint32_t b;                // blocks of the current RSI done
    uint32_t st;               // 0 walking; 1 no coded data set ends inside the input from pos; 2 refused (a run of zero
                               // blocks that overruns its RSI: decode.c:543-544); 3 the walk was cut short (its budget)
};
struct RgEntry {
    uint64_t pos;
    uint32_t b;
    uint32_t live;             // 0: the region has no entry of its own (it belongs to the walk of the region in front)
};

// one coded data set (reference decode.c:402-421 + the block counts of :518-558); with AEC_PAD_RSI the next RSI begins
// on a byte (decode.c:407-408)
template <class P>
AEC_HD void rg_step(P &ps, const Cfg &c, RgState &x)
{
    const uint32_t ref = (x.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
    uint32_t id, nz;
    const uint32_t len = ps.cds(x.pos, ref, id, nz);
    if (!len) {
        x.st = 1u;
        return;
    }
    const uint32_t nb = tr_blocks(c, nz, x.b);
    if (!nb) {
        x.st = 2u;
        return;
    }
    x.pos += len;
    x.b += nb;
    if (x.b >= c.rsi) {
        x.b = 0u;
        if (c.flags & F_PAD_RSI) x.pos = (x.pos + 7u) & ~7ull;
    }
}

// The walk from x to the first RSI start at or behind `target` (the entry of the next region that has one; ~0: to the end
// of the input).  at_rsi(pos) is called at every RSI start in front of it, BEFORE the step; false ends the walk there.
// at_seg(b, pos): at every other coded data set that begins on a multiple of 64 blocks (the segment starts).
template <class P, class FR, class FS>
AEC_HD void rg_walk(P &ps, const Cfg &c, RgState &x, uint64_t target, uint64_t max_bits, FR at_rsi, FS at_seg)
{
    const uint64_t from = x.pos;
    ps.seat(x.pos);
    while (!x.st) {
        if (x.b == 0u) {
            if (x.pos >= target) return;
            if (!at_rsi(x.pos)) return;
        } else if ((x.b & 63u) == 0u) {
            at_seg(x.b, x.pos);
        }
        if (x.pos - from > max_bits) {
            x.st = 3u;
            return;
        }
        rg_step(ps, c, x);
    }
}

// Which regions keep their entries: a guess is taken at most two regions behind its region's first bit, so only the
// NEXT region's can lie in front of it or on it -- then this one is dropped (its stretch belongs to the walk in front).
AEC_HD bool rg_keep(const RgEntry &mine, const RgEntry &next_one, bool have_next)
{
    if (!mine.live) return false;
    return !(have_next && next_one.live && next_one.pos <= mine.pos);
}

}  // namespace aec
