// aec_spec.h -- per-position arithmetic of the SPECULATIVE RSI index (aec_idx.hip: k_spec).
//
// A bare CCSDS 121.0-B-2 stream has no entry points: a coded data set (CDS) is only found by
// parsing its predecessor (reference src/decode.c:402-421).  The walk is serial, but the function
// it iterates -- "a CDS starts at bit q: where does it end?" -- can be tabulated for EVERY bit
// position of a window at once, because with rank/select over the window's 1-bits the end of a
// CDS is a constant number of lookups:
//     split k      : the (bs-ref)-th 1-bit after the header ends the unary part, then (bs-ref)*k bits
//     second ext.  : the (bs/2)-th 1-bit after the header
//     zero blocks  : the first 1-bit after the header (its distance is the run-length code)
//     uncompressed : a constant
// (layouts: reference src/decode.c:462-677, restated in SURVEY.md Appendix A).  On top of that table
// every position of the window's core is tried as the start of an RSI: ref CDS, then table hops
// until rsi blocks are covered.  The results (RSI length per hypothetical start) turn the serial
// walk of the index pass into one table lookup per RSI -- or per window, once the hops have been
// chained inside the window.  Everything here is __host__ __device__ so tests/emul can check it
// against the oracle on the CPU; the product reaches it only through the kernels.
#pragma once

#include "aec_lane.h"

namespace aec {

// Views into the window's LDS arrays (host arrays in the emulator).  All positions are bit offsets
// relative to the window start, bit 0 = MSB of win[0].
struct SpecWin {
    const uint32_t *win;    // stream words in host order; readable up to word (nbits/32 + 1)
    const uint16_t *rank;   // rank[w] = 1-bits in words [0, w); nwords + 1 entries
    const uint16_t *sel;    // sel[m] = word that holds the (32 m + 1)-th 1-bit
    uint32_t nwords;        // words covered by rank/sel
    uint32_t limit;         // bits of the window that belong to the stream (<= 32 nwords)
};

constexpr uint32_t kSpecInvalid = 0xFFFFFFFFu;

// nxt[] entry: bits [0,12) CDS length, bits [12,14) kind
constexpr uint32_t kNxtBlock = 1u << 12;   // one block
constexpr uint32_t kNxtZero = 2u << 12;    // zero-block CDS: run code = len - id_len - 1

AEC_HD uint32_t spec_peek32(const uint32_t *win, uint32_t q)
{
    const uint32_t w = q >> 5, sh = q & 31u;
    const uint64_t two = ((uint64_t)win[w] << 32) | win[w + 1];
    return (uint32_t)((two << sh) >> 32);
}

AEC_HD uint32_t spec_popc(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// 1-bits in [0, q)
AEC_HD uint32_t spec_rank(const SpecWin &s, uint32_t q)
{
    const uint32_t w = q >> 5, sh = q & 31u;
    const uint32_t part = sh ? spec_popc(s.win[w] >> (32u - sh)) : 0u;
    return s.rank[w] + part;
}

// position just behind the j-th 1-bit of the window (j >= 1), kSpecInvalid if there are fewer
AEC_HD uint32_t spec_select(const SpecWin &s, uint32_t j)
{
    // (a window without enough 1-bits -- none at all in a stretch of zero padding -- has no sel[]
    // entry to start from: the search below must not run in that case)
    const bool have = j >= 1u && j <= s.rank[s.nwords];
    if (!have) return kSpecInvalid;
    const uint32_t jj = j;
    uint32_t w = s.sel[(jj - 1u) >> 5];
    if (w >= s.nwords) return kSpecInvalid;          // cannot happen for a consistent table
    // the word is at most 31 one-bits further on: probe four words at once (independent loads),
    // walk on only in sparse regions
    {
        const uint32_t last = s.nwords;
        const uint32_t r1 = s.rank[w + 1 < last ? w + 1 : last], r2 = s.rank[w + 2 < last ? w + 2 : last];
        const uint32_t r3 = s.rank[w + 3 < last ? w + 3 : last], r4 = s.rank[w + 4 < last ? w + 4 : last];
        if (r1 >= jj) {
        } else if (r2 >= jj) {
            w += 1;
        } else if (r3 >= jj) {
            w += 2;
        } else if (r4 >= jj) {
            w += 3;
        } else {
            w += 4;
            while (w + 1u < s.nwords && s.rank[w + 1] < jj) w++;
        }
    }
    uint32_t r = jj - s.rank[w];            // 1-based rank inside the word, from the MSB
    uint32_t x = s.win[w], pos = 0, cnt;
    cnt = spec_popc(x >> 16);
    if (r > cnt) { r -= cnt; pos += 16; x &= 0xFFFFu; } else { x >>= 16; }
    cnt = spec_popc(x >> 8);
    if (r > cnt) { r -= cnt; pos += 8; x &= 0xFFu; } else { x >>= 8; }
    cnt = spec_popc(x >> 4);
    if (r > cnt) { r -= cnt; pos += 4; x &= 0xFu; } else { x >>= 4; }
    cnt = spec_popc(x >> 2);
    if (r > cnt) { r -= cnt; pos += 2; x &= 0x3u; } else { x >>= 2; }
    cnt = x >> 1;
    if (r > cnt) pos += 1;
    return have ? w * 32u + pos + 1u : kSpecInvalid;
}

// Length in bits of a CDS that starts at q (0 = it does not end inside the window's stream bits).
// `run` returns 0 for a CDS of one block, else the zero-block run code fs + 1.
// One code path for all options (the option only selects the operands of ONE rank + select), so
// that lanes of a wave stay together and several positions can be in flight per lane:
//     uncompressed (decode.c:659-677)  constant length, the select result is ignored
//     low entropy  (decode.c:618-644)  header + 1 bit (+ ref); second extension: bs/2 codes,
//                                      zero blocks: 1 code whose length is the run code
//     split k      (decode.c:462-502)  header (+ ref); bs - ref codes, then (bs - ref) * k bits
// 64 window bits from bit q (three words; the caller keeps q + 64 inside the readable words)
AEC_HD uint64_t spec_peek64(const uint32_t *win, uint32_t q)
{
    const uint32_t w = q >> 5, sh = q & 31u;
    const uint64_t a = ((uint64_t)win[w] << 32) | win[w + 1];
    const uint64_t hi = a << sh, lo = sh ? ((uint64_t)win[w + 2] >> (32u - sh)) : 0u;
    return hi | lo;
}

// offset (from the MSB, 0-based) of the n-th 1-bit of x, 1 <= n <= popcount(x)
AEC_HD uint32_t spec_select64(uint64_t x, uint32_t n)
{
    uint32_t pos = 0, cnt;
    uint32_t v = (uint32_t)(x >> 32);
    cnt = spec_popc(v);
    if (n > cnt) { n -= cnt; pos = 32; v = (uint32_t)x; }
    cnt = spec_popc(v >> 16);
    if (n > cnt) { n -= cnt; pos += 16; v &= 0xFFFFu; } else { v >>= 16; }
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

AEC_HD uint32_t spec_cds(const SpecWin &s, const Cfg &c, uint32_t q, uint32_t ref, uint32_t &run)
{
    run = 0;
    const bool in = q + c.id_len + 1u <= s.limit;
    const uint32_t qs = in ? q : 0u;
    const uint32_t h = spec_peek32(s.win, qs);
    const uint32_t id = h >> (32u - c.id_len);
    const bool unc = id == (1u << c.id_len) - 1u, low = id == 0u;
    const uint32_t selbit = (h >> (31u - c.id_len)) & 1u;
    const uint32_t q1 = qs + c.id_len + (low ? 1u : 0u) + (unc ? 0u : ref * c.bps);
    const uint32_t n = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
    const uint32_t add = low ? 0u : n * (id - 1u);
    bool ok = in && (unc || q1 < s.limit);
    const uint32_t q1s = (ok && !unc) ? q1 : 0u;
    // The n-th 1-bit from q1 on ends the unary part.  Short unary regions -- the rule at the optimal
    // k, where the whole region of a block is at most 3n bits -- are resolved inside ONE 64-bit peek
    // (two dependent reads of the window in all); only longer ones take the rank/select tables.
    uint32_t e;
    const bool can_peek = q1s + 64u <= 32u * s.nwords;
    const uint64_t U = can_peek ? spec_peek64(s.win, q1s) : 0u;
    if (can_peek && (uint32_t)
#if defined(__HIP_DEVICE_COMPILE__)
            __popcll(U)
#else
            __builtin_popcountll(U)
#endif
            >= n && n >= 1u)
        e = q1s + spec_select64(U, n) + 1u;
    else
        e = spec_select(s, spec_rank(s, q1s) + n);
    const uint32_t end = unc ? qs + c.id_len + c.bs * c.bps : e + add;
    ok = ok && (unc || e != kSpecInvalid) && end <= s.limit;
    if (!ok) return 0;
    if (low && !selbit) run = e - q1;
    const uint32_t len = end - qs;
    return len < 4096u ? len : 0u;
}

// ---- the common case in straight-line code --------------------------------------------------------
// spec_cds pays for its widest path on every step (~300 instructions per wavefront step on the device: the
// rank / select fallback and its loops are entered as soon as one lane needs them).  spec_cds_fast resolves a
// CDS whose unary part ends inside ONE 64-bit peek -- all of them at the optimal k -- from four consecutive
// window words and nothing else (~100 instructions, no branch); the callers queue the rest for spec_cds.
// Returns the length exactly as spec_cds does, 0 when spec_cds would return 0 for a reason seen here (the CDS
// leaves the window), kSpecUnresolved when the peek does not hold the end of the unary part.
// The window needs 4 readable words behind its last one.
constexpr uint32_t kSpecUnresolved = 0xFFFFFFFFu;

struct SpecQuad {
    uint32_t a, b, c, d;
};

AEC_HD SpecQuad spec_quad(const uint32_t *win, uint32_t w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    struct __attribute__((packed, aligned(4))) Q { uint32_t a, b, c, d; };
    const Q t = *reinterpret_cast<const Q *>(win + w);
    return SpecQuad{t.a, t.b, t.c, t.d};
#else
    return SpecQuad{win[w], win[w + 1], win[w + 2], win[w + 3]};
#endif
}

// upper 32 bits of (a : b) << s, 0 <= s < 32
AEC_HD uint32_t spec_shl_hi(uint32_t a, uint32_t b, uint32_t s)
{
    return (uint32_t)(((((uint64_t)a << 32) | b) << s) >> 32);
}

// REF = 0 / 1: without / with a reference sample; REF = 2: as `ref` says (one code path for both)
template <uint32_t REF>
AEC_HD uint32_t spec_cds_fast(const uint32_t *win, uint32_t limit, const Cfg &c, uint32_t q, uint32_t &run,
                              uint32_t ref = 0)
{
    run = 0;
    if (REF < 2u) ref = REF;
    const uint32_t sh = q & 31u;
    const SpecQuad w = spec_quad(win, q >> 5);
    const uint32_t h = spec_shl_hi(w.a, w.b, sh);
    const uint32_t id = h >> (32u - c.id_len);
    const bool unc = id == (1u << c.id_len) - 1u, low = id == 0u;
    const uint32_t selbit = (h >> (31u - c.id_len)) & 1u;
    const uint32_t off1 = c.id_len + (low ? 1u : 0u) + ((ref && !unc) ? c.bps : 0u);
    const uint32_t n = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
    const uint32_t add = low ? 0u : n * (id - 1u);
    uint32_t hi, lo;
    if (REF) {                                    // behind the reference sample: a second read
        const uint32_t t = q + off1;
        const SpecQuad v = spec_quad(win, t >> 5);
        hi = spec_shl_hi(v.a, v.b, t & 31u);
        lo = spec_shl_hi(v.b, v.c, t & 31u);
    } else {                                      // sh + off1 <= 37: inside the four words
        const uint32_t s1 = sh + off1;
        const bool up = s1 >= 32u;
        const uint32_t a = up ? w.b : w.a, b = up ? w.c : w.b, cc = up ? w.d : w.c;
        hi = spec_shl_hi(a, b, s1 & 31u);
        lo = spec_shl_hi(b, cc, s1 & 31u);
    }
    const uint32_t pc = spec_popc(hi) + spec_popc(lo);
    const uint32_t e = spec_select64(((uint64_t)hi << 32) | lo, n) + 1u;      // (garbage while pc < n)
    const uint32_t len = unc ? c.id_len + c.bs * c.bps : off1 + e + add;
    const bool in = q + c.id_len + 1u <= limit;
    if (in && !unc && pc < n) return kSpecUnresolved;
    if (!in || q + len > limit || len >= 4096u) return 0u;
    if (low && !selbit) run = e;
    return len;
}

AEC_HD uint16_t spec_nxt_entry(const SpecWin &s, const Cfg &c, uint32_t q)
{
    uint32_t run;
    const uint32_t len = spec_cds(s, c, q, 0, run);
    if (!len) return 0;
    return (uint16_t)(len | (run ? kNxtZero : kNxtBlock));
}

// blocks covered by a zero-block CDS with run code nz = fs + 1 at block b of the RSI
// (reference decode.c:524-536); 0 = overruns the RSI (DATA_ERROR in the reference)
AEC_HD uint32_t spec_run_blocks(const Cfg &c, uint32_t nz, uint32_t b)
{
    if (nz == 5u) {
        const uint32_t left_rsi = c.rsi - b, left_seg = 64u - (b % 64u);
        nz = left_rsi < left_seg ? left_rsi : left_seg;
    } else if (nz > 5u) {
        nz--;
    }
    return nz <= c.rsi - b ? nz : 0u;
}

// ---- hop tables: several CDSes per lookup -------------------------------------------------------
// hop entry (u16): bits [0,13) = stream bits covered, bits [13,16) = blocks covered beyond the
// nominal count (zero-block CDSes inside the hop stand for more than one block); 0 = no entry.
// A rest-of-segment run (code 5) never enters a hop table: its block count depends on where in the
// RSI it stands (reference decode.c:528-530), so the walk takes it as a single step.
constexpr uint32_t kHopBitsMask = 0x1FFFu;

AEC_HD uint16_t spec_hop_pack(uint32_t bits, uint32_t extra)
{
    return (bits && bits <= kHopBitsMask && extra <= 7u) ? (uint16_t)(bits | (extra << 13)) : (uint16_t)0;
}

// 4 CDSes from q (hop4), from the per-CDS table.  Straight-line (no early exits): a failed step
// only clears `ok` and the position is clamped, so that several positions can be in flight per lane.
AEC_HD uint16_t spec_hop4(const uint16_t *nxt, const Cfg &c, uint32_t limit, uint32_t q)
{
    uint32_t pos = q, extra = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ok = ok && pos < limit;
        const uint32_t e = nxt[ok ? pos : q];
        const uint32_t len = e & 0xFFFu;
        const uint32_t code = len - c.id_len - 1u;
        const bool zero = e & kNxtZero;
        ok = ok && e != 0u && !(zero && code == 5u);
        extra += zero ? (code < 5u ? code : code - 1u) - 1u : 0u;
        pos += len;
    }
    return ok ? spec_hop_pack(pos - q, extra) : (uint16_t)0;
}

// 16 CDSes from q (hop16), from the hop4 table
AEC_HD uint16_t spec_hop16(const uint16_t *hop4, uint32_t limit, uint32_t q)
{
    uint32_t pos = q, extra = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ok = ok && pos < limit;
        const uint32_t e = hop4[ok ? pos : q];
        ok = ok && e != 0u;
        pos += e & kHopBitsMask;
        extra += e >> 13;
    }
    return ok ? spec_hop_pack(pos - q, extra) : (uint16_t)0;
}

// First CDS of an RSI at q (carries the reference sample when the preprocessor is on), in the
// nxt[] entry format; the run code of a zero-block CDS is len - id_len - 1 - (pp ? bps : 0).
AEC_HD uint16_t spec_first_entry(const SpecWin &s, const Cfg &c, uint32_t q)
{
    uint32_t run;
    const uint32_t len = spec_cds(s, c, q, (c.flags & F_PREPROCESS) ? 1u : 0u, run);
    if (!len) return 0;
    return (uint16_t)(len | (run ? kNxtZero : kNxtBlock));
}

// State of one RSI walk: p + walked bits = pos, b = blocks covered so far.
// spec_walk_init consumes the first CDS; false = unresolved.
AEC_HD bool spec_walk_init(const Cfg &c, uint32_t first_entry, uint32_t p, uint32_t &pos, uint32_t &b)
{
    if (!first_entry) return false;
    const uint32_t len = first_entry & 0xFFFu;
    uint32_t n = 1;
    if (first_entry & kNxtZero) {
        const uint32_t code = len - c.id_len - 1u - ((c.flags & F_PREPROCESS) ? c.bps : 0u);
        n = spec_run_blocks(c, code, 0);
        if (!n) return false;
    }
    pos = p + len;
    b = n;
    return true;
}

// One step of the walk: the widest table entry that fits the blocks left.  The three entries are
// read up front (independent loads, one LDS latency).  false = unresolved.
AEC_HD bool spec_walk_step(const Cfg &c, const uint16_t *nxt, const uint16_t *hop4, const uint16_t *hop16,
                           uint32_t limit, uint32_t &pos, uint32_t &b)
{
    if (pos >= limit) return false;
    const uint32_t e16 = hop16 ? hop16[pos] : 0u, e4 = hop4 ? hop4[pos] : 0u, e1 = nxt[pos];
    const uint32_t left = c.rsi - b;
    if (e16 && 16u + (e16 >> 13) <= left) {
        pos += e16 & kHopBitsMask;
        b += 16u + (e16 >> 13);
        return true;
    }
    if (e4 && 4u + (e4 >> 13) <= left) {
        pos += e4 & kHopBitsMask;
        b += 4u + (e4 >> 13);
        return true;
    }
    if (!e1) return false;
    const uint32_t len = e1 & 0xFFFu;
    uint32_t n = 1;
    if (e1 & kNxtZero) {
        n = spec_run_blocks(c, len - c.id_len - 1u, b);
        if (!n) return false;
    }
    pos += len;
    b += n;
    return true;
}

// Length in bits of a whole RSI (c.rsi blocks) that starts at p; 0 = not resolved inside the
// window (walk leaves it, meets an invalid entry, or a zero run overruns the RSI).
AEC_HD uint32_t spec_rsi(const SpecWin &s, const Cfg &c, const uint16_t *nxt, const uint16_t *hop4,
                         const uint16_t *hop16, uint32_t p)
{
    uint32_t pos, b;
    if (!spec_walk_init(c, spec_first_entry(s, c, p), p, pos, b)) return 0;
    while (b < c.rsi)
        if (!spec_walk_step(c, nxt, hop4, hop16, s.limit, pos, b)) return 0;
    return pos - p;
}

}  // namespace aec
