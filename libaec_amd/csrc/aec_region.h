// aec_region.h -- per-lane arithmetic of the REGION index (aec_region.hip; DESIGN.md section 2, scheme "regions").
//
// A bare stream of coded data sets has no entry points (reference src/decode.c:402-421), but a lane that stands on an
// RSI start walks the stream from there at a couple of hundred instructions per coded data set, and a chip full of such
// lanes passes over a gigabyte in a millisecond or two.  What a lane in the middle of the stream lacks is the place to
// stand on.  The stream goes in regions; every region but the first GUESSES its entry -- the first RSI start behind its
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
//      coded data sets.  The walk goes on for kRgDefer more (a jump of the data: the sum recovers, nothing happened);
//   3. TEST the boundary in front of the first option that did not fit, and the two in front of that (the first garbage
//      options may have fitted by chance), as RSI starts: the coded data set parsed WITH a reference sample and
//      kRgTest more, scored the same way, then the same chain WITHOUT the reference sample -- an RSI start is where the
//      first scores kRgTestPass and beats the second by kRgMargin (at any other boundary the plain chain IS the true
//      one and scores no less).  The best candidate wins;
//   4. VERIFY (RSIs of up to kRgVerifyMaxRsi blocks): the candidate's whole RSI is walked with its bookkeeping and the
//      same test must hold where it ends -- a wrong candidate's RSI ends on no RSI start.
// None: the data made a jump (on with the walk), or -- three times in a row -- the chain is lost (a new anchor).
// An RSI start that the plain parse survives (an uncompressed first block holds its reference sample as sample 0,
// decode.c:659-677; a lucky landing) is simply passed: the next one is rsi blocks on.
// The guess is a state machine that takes ONE coded data set per step whatever it is doing: the lanes of a wavefront
// are in different states, and this way they share the parse.
//
// Positions in the inner loops are 32-bit offsets from the lane's base (RgRing::base_bits); a walk that would leave
// their range is cut short (RgState::st 3) and nothing is delivered.
//
// Everything here is __host__ __device__: tests/emul/region_emul.cpp runs the same functions on the CPU against the
// RSI starts the oracle's encoder reports.
#pragma once

#include "aec_trunk.h"

namespace aec {

constexpr int32_t kRgAnchorScore = 16, kRgAnchorFail = -4;
constexpr uint32_t kRgHist = 12;           // boundaries the walk remembers (RgGuess::hp0 .. hp11)
constexpr uint32_t kRgCand = 3;            // ... of which three are tried as the RSI start
constexpr uint32_t kRgDefer = 3;           // coded data sets the walk goes on before it tests
constexpr uint32_t kRgTest = 16;           // coded data sets behind a candidate's first
constexpr int32_t kRgTestPass = 18, kRgTestFail = -8;
constexpr int32_t kRgMargin = 8;           // ... and by how much the chain WITH a reference sample must beat the one without
constexpr uint32_t kRgVerifyMaxRsi = 512;  // RSIs of up to this many blocks: the candidate's RSI is walked and its END tested too
constexpr int32_t kRgTestPass2 = 12, kRgMargin2 = 0;
constexpr int32_t kRgLeakStart = 32, kRgSuspect = -8, kRgHealthy = 12;    // leaky sum, in quarters
constexpr uint32_t kRgRelMax = 0xF0000000u;                               // 32-bit positions: as far as a walk goes from its base

// score of an option `id` behind the option `prev`: near log2 of P(distance | true chain) / P(distance | random bits)
AEC_HD int32_t rg_score(uint32_t id, uint32_t prev, uint32_t id_len)
{
    const uint32_t d = id > prev ? id - prev : prev - id;
    if (d <= 1u) return 2;
    if (id_len >= 5u) return d == 2u ? 1 : (d == 3u ? -2 : -6);
    return d == 2u ? -2 : -6;
}

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

// ---- the lane's ring ------------------------------------------------------------------------------------------------
// A lane that follows a chain through device memory waits for memory at every coded data set, and a wavefront waits for
// whichever of its 64 lanes touched a new line of HBM: microseconds per step.  Here a lane keeps kRgRingWords words of
// its stretch of the stream in a ring (LDS on the device: word i of the stream at ring[slot * stride], a column per lane,
// conflict-free; the first kRgRingMirror slots are mirrored behind the last so that the seven words of a parse are
// seven consecutive rows), parses out of it, and the WAVEFRONT tops the rings up together every `period` steps: up to
// eight 16-byte loads per lane leave, and are stored at the next top-up, `period` steps -- a microsecond or more --
// later: what memory takes is over by then for every lane.  (Measured: topping up at once whenever a lane runs short,
// with the wait for memory there, costs the wavefront a third more -- the lanes consume at different rates and every
// few steps one of them is short.)  The ring keeps kRgRingBack words behind the parse (the guess goes back a few coded
// data sets for its tests); a lane that jumps out of its ring, or consumed faster than the top-ups bring, fills up at
// once and the wavefront waits.
constexpr uint32_t kRgRingMirror = 8, kRgRingChunks = 8;
// (WORDS: the ring's size, 32 or 64 -- the LDS it takes decides how many wavefronts a CU holds, and the walks of short
// coded data sets, which go forward only, do with the smaller; a quarter of it stays behind the parse)
constexpr uint32_t rg_ring_rows(uint32_t words) { return words + kRgRingMirror; }

struct RgChunk {
    uint32_t a, b, c, d;      // four stream words as they lie in memory
};

// STRIDE: words between two rows of a lane's column (64 in LDS: a row holds the 64 lanes' words; 1 in the emulator)
template <uint32_t STRIDE, uint32_t WORDS = 64u>
struct RgRingT {
    static constexpr uint32_t stride = STRIDE, kRgRingWords = WORDS, kRgRingBack = WORDS / 4u;
    static_assert(WORDS == 32u || WORDS == 64u, "a power of two that holds a parse's seven words and the top-up's chunks");
    const TrStream &s;
    const Cfg &c;
    uint32_t *ring;
    uint64_t base_bits;        // positions are base_bits + rel; a multiple of 128
    uint32_t wbase;            // low bits of base_bits / 32 (the slot of relative word 0)
    uint32_t lo, hi;           // relative words [lo, hi) are in the ring; multiples of 4, hi - lo <= kRgRingWords
    uint32_t end_rel;          // end_bit - base_bits, clipped to 32 bits
    bool aligned16;            // the stream's buffer begins on a 16-byte boundary (a chunk is one load)
    uint32_t period, tick;     // steps between the wavefront's top-ups
    uint32_t pv;               // chunks in flight: relative words [hi, hi + 4 pv)
    RgChunk p0, p1, p2, p3, p4, p5, p6, p7;      // (named: see RgGuess::hp0)

    AEC_HD void init(uint32_t *ring_, uint32_t period_)
    {
        ring = ring_;
        period = period_ ? period_ : 1u;
        aligned16 = (reinterpret_cast<uintptr_t>(s.words) & 15u) == 0u;
        seat(0);
    }
    // positions from `pos` on (and a little in front of it) become addressable
    AEC_HD void seat(uint64_t pos)
    {
        base_bits = pos & ~127ull;
        wbase = (uint32_t)(base_bits >> 5);
        lo = hi = 0;
        pv = 0;
        tick = 0;
        const uint64_t left = s.end_bit > base_bits ? s.end_bit - base_bits : 0u;
        end_rel = left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    }
    AEC_HD uint32_t rel_of(uint64_t pos) const { return (uint32_t)(pos - base_bits); }
    AEC_HD uint64_t pos_of(uint32_t rel) const { return base_bits + rel; }

    AEC_HD RgChunk load_chunk(uint32_t rw) const
    {
        const uint64_t w = (base_bits >> 5) + rw;
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
    AEC_HD void put(uint32_t rw, const RgChunk &v)
    {
        const uint32_t row = (wbase + rw) & (kRgRingWords - 1u);      // (a multiple of 4: the chunk does not wrap)
        uint32_t *q = ring + (size_t)row * stride;
        const uint32_t x = bswap32(v.a), y = bswap32(v.b), z = bswap32(v.c), w = bswap32(v.d);
        q[0] = x;
        q[stride] = y;
        q[2u * stride] = z;
        q[3u * stride] = w;
        if (row < kRgRingMirror) {
            uint32_t *m = q + (size_t)kRgRingWords * stride;
            m[0] = x;
            m[stride] = y;
            m[2u * stride] = z;
            m[3u * stride] = w;
        }
    }
    // the chunks in flight into the ring
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
    // room is made behind the parse at relative word rw, up to eight chunks leave
    AEC_HD void issue(uint32_t rw)
    {
        const uint32_t keep = (rw > kRgRingBack ? rw - kRgRingBack : 0u) & ~3u;
        if (keep > lo) lo = keep < hi ? keep : hi;
        const uint32_t room = (kRgRingWords - (hi - lo)) >> 2;
        const uint32_t n = room < kRgRingChunks ? room : kRgRingChunks;
        pv = n;
#if defined(__HIP_DEVICE_COMPILE__)
        // (all of them inside the buffer, which begins on a 16-byte boundary -- the rule: one address, eight loads)
        const uint64_t w = (base_bits >> 5) + hi;
        if (aligned16 && w + 4u * n <= s.nwords) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s.words + w);
            if (n > 0u) { const uint4 v = src[0]; p0 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 1u) { const uint4 v = src[1]; p1 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 2u) { const uint4 v = src[2]; p2 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 3u) { const uint4 v = src[3]; p3 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 4u) { const uint4 v = src[4]; p4 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 5u) { const uint4 v = src[5]; p5 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 6u) { const uint4 v = src[6]; p6 = RgChunk{v.x, v.y, v.z, v.w}; }
            if (n > 7u) { const uint4 v = src[7]; p7 = RgChunk{v.x, v.y, v.z, v.w}; }
            return;
        }
#endif
        if (n > 0u) p0 = load_chunk(hi);
        if (n > 1u) p1 = load_chunk(hi + 4u);
        if (n > 2u) p2 = load_chunk(hi + 8u);
        if (n > 3u) p3 = load_chunk(hi + 12u);
        if (n > 4u) p4 = load_chunk(hi + 16u);
        if (n > 5u) p5 = load_chunk(hi + 20u);
        if (n > 6u) p6 = load_chunk(hi + 24u);
        if (n > 7u) p7 = load_chunk(hi + 28u);
    }

    // The coded data set at relative position rel, parsed without (ref 0) or with (1) a reference sample: its length in
    // bits (0: none ends inside the stream), option id and zero-run code -- what tr_cds(s, c, base_bits + rel, ref, nz)
    // returns, bit for bit (tests/emul/region_emul.cpp compares them at every bit of whole streams).  32-bit arithmetic
    // throughout: this is the inner loop of every pass of the scheme.
    AEC_HD uint32_t cds(uint32_t rel, uint32_t ref, uint32_t &id, uint32_t &nz)
    {
        nz = 0;
        id = 0;
        const bool in = rel < end_rel && end_rel - rel > c.id_len;
        const uint32_t rw = rel >> 5;
        // the wavefront's top-up (its lanes step together): what left at the last one is stored, the next loads leave
        if (++tick >= period) {
            tick = 0;
            if (in) {
                land();
                issue(rw);
            }
        }
        // a lane whose seven words are not in its ring: it jumped out of it (and of what is on its way), or consumed
        // faster than the top-ups bring -- filled at once, the wavefront waits
        const uint32_t dl = rw - lo, span = hi - lo;                  // (behind lo: dl is huge)
        const bool shortage = in && (dl > span || dl + 7u > span);
        if (AEC_ANY(shortage)) {
            if (shortage) {
                if (rw < lo || rw > hi + 4u * pv) {
                    pv = 0;
                    lo = hi = rw & ~3u;                               // out of the ring: anew from here
                }
                do {
                    land();
                    issue(rw);
                } while (rw + 7u > hi);
            }
        }
        if (!in) return 0;
        const uint32_t *q = ring + (size_t)((wbase + rw) & (kRgRingWords - 1u)) * stride;
        const uint32_t w0 = q[0], w1 = q[stride];
        const uint32_t left = end_rel - rel;
        const uint32_t sh = rel & 31u;
        const uint32_t h = spec_shl_hi(w0, w1, sh);
        id = h >> (32u - c.id_len);
        const bool unc = id == (1u << c.id_len) - 1u, low = id == 0u;
        const uint32_t selbit = (h >> (31u - c.id_len)) & 1u;
        const uint32_t hdr = c.id_len + (low ? 1u : 0u) + ref * c.bps;          // <= 38
        const uint32_t need = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
        const uint32_t add = low ? 0u : need * (id - 1u);
        const uint32_t o2 = sh + hdr, i = o2 >> 5, t2 = o2 & 31u;               // o2 < 70: i <= 2
        // (the five words the unary part begins in: a second read at their own row -- cheaper than picking them out of
        // seven by i; rows up to 6 behind the first: the mirror rows serve them)
        const uint32_t *qa = q + (size_t)i * stride;
        const uint32_t a0 = qa[0], a1 = qa[stride], a2 = qa[2u * stride], a3 = qa[3u * stride], a4 = qa[4u * stride];
        // 128 bits of unary part in four pieces; their running counts of 1-bits say which holds the need-th
        const uint32_t u0 = spec_shl_hi(a0, a1, t2), u1 = spec_shl_hi(a1, a2, t2), u2 = spec_shl_hi(a2, a3, t2),
                       u3 = spec_shl_hi(a3, a4, t2);
        const uint32_t c0 = spec_popc(u0), c1 = c0 + spec_popc(u1), c2 = c1 + spec_popc(u2), c3 = c2 + spec_popc(u3);
        const bool far = !unc && c3 < need;
        if (AEC_ANY(far)) {                                                     // (rare: from memory)
            if (far) return tr_cds(s, c, base_bits + rel, ref, nz);
        }
        const uint32_t k = (need > c0 ? 1u : 0u) + (need > c1 ? 1u : 0u) + (need > c2 ? 1u : 0u);
        const uint32_t u = k == 0u ? u0 : (k == 1u ? u1 : (k == 2u ? u2 : u3));
        const uint32_t before = k == 0u ? 0u : (k == 1u ? c0 : (k == 2u ? c1 : c2));
        const uint32_t used = hdr + 32u * k + rg_select32(u, need - before) + 1u;
        if (low && !selbit) nz = used - hdr;
        const uint32_t len = unc ? c.id_len + c.bs * c.bps : used + add;
        return len <= left ? len : 0u;
    }
};

// steps between the top-ups of the rings: so that eight chunks (128 bytes) are twice what a lane consumes in between
AEC_HD uint32_t rg_ring_period(uint64_t avg_cds_bits)
{
    const uint64_t p = avg_cds_bits ? (128u * 8u) / (2u * avg_cds_bits) : 8u;
    return p < 1u ? 1u : (p > 8u ? 8u : (uint32_t)p);
}

struct RgMemParser {                       // straight from memory (what the ring's parse is checked against)
    const TrStream &s;
    const Cfg &c;
    AEC_HD uint32_t cds(uint64_t q, uint32_t ref, uint32_t &id, uint32_t &nz) const
    {
        nz = 0;
        id = 0;
        if (q + c.id_len >= s.end_bit) return 0;
        id = (uint32_t)(tr_peek64(s, q) >> (64u - c.id_len));
        return tr_cds(s, c, q, ref, nz);
    }
};

// ---- the guess -------------------------------------------------------------------------------------------------------
struct RgGuess {
    enum : uint32_t { ANCHOR = 0, WALK = 1, TEST = 2, VERIFY = 3, FOUND = 4, NONE = 5 };
    uint32_t mode;
    uint32_t q;                // where the next parse begins (positions: relative to the ring's base)
    uint32_t ref;              // ... with a reference sample
    uint32_t prev;             // option in front of q; bit 8: that coded data set was a run of zero blocks
    int32_t S;                 // score of the chain at hand (anchor, test)
    uint32_t steps;
    uint32_t t;                // anchor: first bit of the chain at hand
    // walk: the last boundaries (hp0 the latest), ~0 = none; and `prev` in front of each
    // (named, not arrays: the compiler keeps a struct with arrays in scratch memory)
    uint32_t hp0, hp1, hp2, hp3, hp4, hp5, hp6, hp7, hp8, hp9, hp10, hp11;
    uint32_t ho0, ho1, ho2, ho3, ho4, ho5, ho6, ho7, ho8, ho9, ho10, ho11;
    int32_t L;                 // leaky sum of the walk's scores
    uint32_t fails;            // tests in a row that found nothing
    uint32_t defer;            // walk: coded data sets until the test (0: no suspicion)
    uint32_t age;              // walk: coded data sets since the first option of this stretch that did not fit, that one included
    uint32_t j, jend;          // test: candidate at hand, one past the last
    uint32_t sub;              // test: 0 = the chain with a reference sample, 1 = the one without from the same boundary
    int32_t Sg;                // test: score of the first of the two
    int32_t best;              // test: best score of a candidate so far, and which
    uint32_t best_j;
    uint32_t wq;               // test: where the walk goes on
    uint32_t wprev;
    uint32_t vb;               // verify: blocks of the candidate's RSI done
    uint32_t second;           // test: 1 = at the END of the candidate's RSI (the next RSI start, if the candidate is one)
    uint32_t cq;               // test: the boundary under test
    uint32_t cprev;
    uint32_t found;            // FOUND: the RSI start
    uint32_t parses;

    AEC_HD void init(uint32_t from)
    {
        mode = ANCHOR;
        q = t = from;
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
        j = jend = 0;
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
    AEC_HD void new_anchor(uint32_t at, uint32_t end_rel, const Cfg &c)
    {
        mode = (at < end_rel && end_rel - at > c.id_len) ? ANCHOR : NONE;
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
    AEC_HD void resume_walk(uint32_t end_rel, const Cfg &c)
    {
        fails++;
        if (fails >= 3u) {
            new_anchor(wq + 1u, end_rel, c);
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
    AEC_HD void test_begin(uint32_t at, uint32_t opt_in_front)
    {
        mode = TEST;
        cq = q = at;
        cprev = prev = opt_in_front;
        ref = 1;
        sub = 0;
        S = 0;
        steps = 0;
    }
    AEC_HD void next_candidate(uint32_t end_rel, const Cfg &c)
    {
        j++;
        const uint32_t p = (j < kRgHist && j < jend) ? hist_pos(j) : ~0u;
        if (p != ~0u) {
            test_begin(p, hist_opt(j));
            return;
        }
        // all tried
        if (best < kRgTestPass) {
            resume_walk(end_rel, c);
            return;
        }
        found = hist_pos(best_j);
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
    AEC_HD void step(const Cfg &c, uint32_t end_rel, uint32_t len, uint32_t id, uint32_t nz)
    {
        parses++;
        if (mode == ANCHOR) {
            uint32_t pv = steps ? prev : id;             // (the first has nothing in front of it)
            const int32_t sc = score(c, len, id, nz, pv);
            S += sc;
            if (!len || S <= kRgAnchorFail) {
                new_anchor(t + 1u, end_rel, c);
                return;
            }
            prev = pv;
            steps++;
            q += len;
            if (S >= kRgAnchorScore) {
                mode = WALK;
                L = kRgLeakStart;
                fails = 0;
                hist_clear();
                defer = 0;
                age = 0;
            }
            return;
        }
        if (mode == WALK) {
            if (!len) {
                new_anchor(q + 1u, end_rel, c);
                return;
            }
            const uint32_t was = prev;
            const int32_t sc = score(c, len, id, nz, prev);
            L = L - (L >> 2) + 4 * sc;
            if (L >= kRgHealthy) fails = 0;
            age = age ? age + 1u : (sc < 0 ? 1u : 0u);
            if (L >= kRgHealthy && sc > 0) age = 0u;
            hist_push(q, was);
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
                    next_candidate(end_rel, c);
                }
            } else if (L < kRgSuspect) {
                defer = kRgDefer;
            }
            return;
        }
        if (mode == VERIFY) {
            const uint32_t nb = len ? tr_blocks(c, nz, vb) : 0u;
            if (!nb) {
                resume_walk(end_rel, c);
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
                    resume_walk(end_rel, c);
                else
                    next_candidate(end_rel, c);  // (ties between candidates: the later boundary)
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
                    resume_walk(end_rel, c);
                return;
            }
            if (ok) {
                best = Sg;
                best_j = j;
            }
            next_candidate(end_rel, c);
            return;
        }
        steps++;
        q += len;
        ref = 0;
    }
};

// The first RSI start the guess from bit `from` recognises within `budget` coded data sets, in front of `limit` bits
// from there (the walk of the guess does not go beyond).
template <class RING>
AEC_HD bool rg_guess(RING &ps, const Cfg &c, uint64_t from, uint32_t limit, uint32_t budget, uint64_t &rsi_start,
                     uint64_t *parses = nullptr)
{
    ps.seat(from);
    const uint32_t r0 = ps.rel_of(from);
    RgGuess g;
    g.init(r0);
    if (!(r0 < ps.end_rel && ps.end_rel - r0 > c.id_len)) g.mode = RgGuess::NONE;
    while (g.busy() && g.parses < budget && !(g.mode <= RgGuess::WALK && g.q - r0 >= limit)) {
        uint32_t id, nz;
        const uint32_t len = ps.cds(g.q, g.ref, id, nz);
        g.step(c, ps.end_rel, len, id, nz);
    }
    if (parses) *parses += g.parses;
    if (g.mode != RgGuess::FOUND) return false;
    rsi_start = ps.pos_of(g.found);
    return true;
}

// ---- the walk with the RSI's bookkeeping ----------------------------------------------------------------------------
struct RgState {
    uint64_t pos;
    uint32_t b;                // blocks of the current RSI done
    uint32_t st;               // 0 walking; 1 no coded data set ends inside the input from pos; 2 refused (a run of zero
                               // blocks that overruns its RSI: decode.c:543-544); 3 the walk was cut short (its bound)
};
struct RgEntry {
    uint64_t pos;
    uint32_t b;
    uint32_t live;             // 0: the region has no entry of its own (it belongs to the walk of the region in front)
};

// The walk from x to the first RSI start at or behind `target` (the entry of the next region that has one; ~0: to the end
// of the input), one coded data set per step (reference decode.c:402-421 + the block counts of :518-558; with
// AEC_PAD_RSI the next RSI begins on a byte, decode.c:407-408).  at_rsi(pos) is called at every RSI start in front of
// the target, BEFORE the step; false ends the walk there.  at_seg(b, pos): at every other coded data set that begins on
// a multiple of 64 blocks (the segment starts).  Cut short (st 3) after max_bits.
template <class RING, class FR, class FS>
AEC_HD void rg_walk(RING &ps, const Cfg &c, RgState &x, uint64_t target, uint64_t max_bits, FR at_rsi, FS at_seg)
{
    ps.seat(x.pos);
    const uint64_t base = ps.base_bits;
    uint32_t rel = ps.rel_of(x.pos), b = x.b, st = x.st;
    const uint64_t tr64 = target > base ? target - base : 0u;
    const uint32_t trel = tr64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tr64;
    const uint64_t mr64 = max_bits > (uint64_t)kRgRelMax ? (uint64_t)kRgRelMax : max_bits + rel;
    const uint32_t mrel = mr64 > kRgRelMax ? kRgRelMax : (uint32_t)mr64;
    const bool pp = (c.flags & F_PREPROCESS) != 0u, pad = (c.flags & F_PAD_RSI) != 0u;
    while (!st) {
        if (b == 0u) {
            if (rel >= trel) break;
            if (!at_rsi(base + rel)) break;
        } else if ((b & 63u) == 0u) {
            at_seg(b, base + rel);
        }
        if (rel > mrel) {
            st = 3u;
            break;
        }
        uint32_t id, nz;
        const uint32_t len = ps.cds(rel, (b == 0u && pp) ? 1u : 0u, id, nz);
        const uint32_t nb = len ? tr_blocks(c, nz, b) : 0u;
        if (!nb) {
            st = len ? 2u : 1u;
            break;
        }
        rel += len;
        b += nb;
        if (b >= c.rsi) {
            b = 0u;
            if (pad) rel = (rel + 7u) & ~7u;             // (base is a multiple of 8)
        }
    }
    x.pos = base + rel;
    x.b = b;
    x.st = st;
}

// Which regions keep their entries: a guess is taken at most two regions behind its region's first bit, so only the
// NEXT region's can lie in front of it or on it -- then this one is dropped (its stretch belongs to the walk in front).
AEC_HD bool rg_keep(const RgEntry &mine, const RgEntry &next_one, bool have_next)
{
    if (!mine.live) return false;
    return !(have_next && next_one.live && next_one.pos <= mine.pos);
}

}  // namespace aec
