// aec_spec2.h -- SPARSE speculative RSI index: per-window arithmetic (aec_idx.hip: k_spec2).
//
// aec_spec.h gives, with rank/select over a window's 1-bits, "a coded data set (CDS) starts at bit q:
// where does it end?" in O(1).  The first index pass tabulated that -- and the RSI hypothesis on top
// of it -- for EVERY bit position of every window and threw nearly all of it away.  This version only
// works where a CDS can actually start:
//
//   1. SYNC CHAINS.  The CDS parse self-synchronises like any prefix code: a walk that starts at an
//      arbitrary bit falls onto the true chain of CDS boundaries after a few codes.  A lane every
//      `stride` bits burns in `burn` codes and then MARKS the boundaries it visits in a bitmap until
//      it reaches one that is marked already (from there on another lane covers the chain).  The
//      marked positions -- candidates -- are the true boundaries plus a few bogus ones.
//   2. TABLES ON CANDIDATES ONLY: CDS length, 4- and 16-CDS hops (through marked positions), then the
//      unit hypothesis "an RSI (short RSIs) / a first or an inner segment of 64 blocks (long RSIs)
//      starts here" as a walk over those hops; a walk that leaves the marked chain (the reference
//      sample of a hypothetical RSI start shifts it) computes the CDS ends on demand until it is back.
//   3. CHAINS of units inside the window core: exit position + units covered, per candidate.
//
// Exactness: every table entry is the exact function value at its position (marking only selects
// WHERE entries exist); the walkers read entries at true unit starts only and fall back to the serial
// CDS walk where an entry is missing, so results never depend on the speculation.  Everything is
// __host__ __device__: tests/emul runs it window by window on the CPU against the oracle.
#pragma once

#include "aec_spec.h"

namespace aec {

constexpr uint32_t kS2NoIndex = 0xFFFFFFFFu;
constexpr uint16_t kS2NoSucc = 0xFFFFu;
constexpr uint32_t kS2Limited = 0xFFFFu;          // k_spec2's note in ub[]: the walk left the window (see k_bridge)
constexpr uint32_t kS2HopIdxMask = 0x1FFFu;      // (a window has at most 8192 candidates)

// Window-relative views (LDS on the device, host arrays in the emulator).  Bit q of the window is bit
// (31 - q % 32) of word q / 32, for the stream words and for the mark bitmap alike.
struct S2Win {
    SpecWin s;               // stream words, rank/select over them, valid bits
    const uint32_t *marks;   // candidate bitmap, s.nwords words
    const uint16_t *mpre;    // mpre[w] = candidates in words [0, w)
    const uint16_t *cnxt;    // per candidate: CDS entry as in aec_spec.h nxt[] (len | kind), 0 = none
    const uint16_t *csucc;   // per candidate: index of the candidate its CDS ends at, kS2NoSucc = none (unmarked / outside)
    const uint16_t *chop4;   // per candidate: 4 CDSes on -- bits [0,13) INDEX of the candidate reached (never 0: a hop
                             // goes forward), bits [13,16) blocks covered beyond the nominal count; 0 = none
    const uint16_t *chop16;  // per candidate: 16 CDSes on, same format
    const uint16_t *cpos;    // per candidate: its bit position
    uint32_t ncand;
};

AEC_HD bool s2_marked(const uint32_t *marks, uint32_t q) { return (marks[q >> 5] >> (31u - (q & 31u))) & 1u; }

// index of the candidate at bit q, kS2NoIndex if q is not marked
AEC_HD uint32_t s2_index(const S2Win &w, uint32_t q)
{
    const uint32_t word = w.marks[q >> 5], sh = q & 31u;
    if (!((word >> (31u - sh)) & 1u)) return kS2NoIndex;
    return (uint32_t)w.mpre[q >> 5] + (sh ? spec_popc(word >> (32u - sh)) : 0u);
}

// One step of a sync chain: the CDS at q parsed WITHOUT a reference sample; 0 = no CDS ends inside
// the window from here.
AEC_HD uint32_t s2_chain_step(const SpecWin &s, const Cfg &c, uint32_t q)
{
    if (q >= s.limit) return 0;
    uint32_t run;
    return spec_cds(s, c, q, 0, run);
}

#if !defined(__HIPCC__) && defined(AEC_S2_COUNT)
#define S2_COUNT(i) (s2_counters[i]++)
static unsigned long long s2_counters[8];
#else
#define S2_COUNT(i) ((void)0)
#endif
#ifndef S2_NOTE
#define S2_NOTE(pos) ((void)0)      // (emulator statistics: positions parsed on demand)
#endif
#ifndef S2_HIST
#define S2_HIST(parses, steps) ((void)0)   // (emulator statistics: on-demand parses and table steps of a unit walk)
#endif

// Tables linked by candidate INDEX: a table step is then one read (no search for the candidate at a position),
// and a hop is not limited by the bits it covers.
AEC_HD uint16_t s2_hop_pack(uint32_t idx, uint32_t extra)
{
    return (idx != 0u && idx <= kS2HopIdxMask && extra <= 7u) ? (uint16_t)(idx | (extra << 13)) : (uint16_t)0;
}

// successor of candidate idx through its own CDS
AEC_HD uint16_t s2_succ(const S2Win &w, uint32_t idx)
{
    const uint32_t e = w.cnxt[idx];
    if (!e) return kS2NoSucc;
    const uint32_t pos = (uint32_t)w.cpos[idx] + (e & 0xFFFu);
    if (pos >= w.s.limit) return kS2NoSucc;
    const uint32_t j = s2_index(w, pos);
    return j == kS2NoIndex ? kS2NoSucc : (uint16_t)j;
}

// hop over 4 CDSes from candidate `idx` through MARKED positions (0 = leaves the marked chain, a
// rest-of-segment run inside, or more than 7 extra blocks)
AEC_HD uint16_t s2_hop4(const S2Win &w, const Cfg &c, uint32_t idx)
{
    uint32_t extra = 0, i = idx;
    for (int k = 0; k < 4; k++) {
        const uint32_t e = w.cnxt[i];
        const uint32_t len = e & 0xFFFu, code = len - c.id_len - 1u;
        const bool zero = e & kNxtZero;
        if (!e || (zero && code == 5u)) return 0;
        extra += zero ? (code < 5u ? code : code - 1u) - 1u : 0u;
        const uint32_t nx = w.csucc[i];
        if (nx == kS2NoSucc) return 0;
        i = nx;
    }
    return s2_hop_pack(i, extra);
}

AEC_HD uint16_t s2_hop16(const S2Win &w, uint32_t idx)
{
    uint32_t extra = 0, i = idx;
    for (int k = 0; k < 4; k++) {
        const uint32_t e = w.chop4[i];
        if (!e) return 0;
        extra += e >> 13;
        i = e & kS2HopIdxMask;
    }
    return s2_hop_pack(i, extra);
}

// One table step of a unit walk standing on candidate `idx` with `b` blocks of the RSI done, never beyond `bend`:
// the widest table entry that fits.  Returns 1 = went on (idx, b updated), 2 = the unit is complete, `end` is
// where it ends (a last single CDS may end on an unmarked position), 0 = unresolved because the walk leaves the
// WINDOW, 3 = not an RSI (a zero run overrunning the unit), 4 = unresolved because it leaves the marked chain.
AEC_HD uint32_t s2_table_step(const S2Win &w, const Cfg &c, uint32_t &idx, uint32_t &b, uint32_t bend, uint32_t &end)
{
    const uint32_t left = bend - b;
    const uint32_t e16 = w.chop16[idx], e4 = w.chop4[idx], e1 = w.cnxt[idx];
    if (e16 && 16u + (e16 >> 13) <= left) {
        S2_COUNT(0);
        b += 16u + (e16 >> 13);
        idx = e16 & kS2HopIdxMask;
    } else if (e4 && 4u + (e4 >> 13) <= left) {
        S2_COUNT(1);
        b += 4u + (e4 >> 13);
        idx = e4 & kS2HopIdxMask;
    } else {
        S2_COUNT(2);
        if (!e1) return (uint32_t)w.cpos[idx] + 4096u > w.s.limit ? 0u : 4u;
        const uint32_t len = e1 & 0xFFFu;
        uint32_t n = 1;
        if (e1 & kNxtZero) {
            n = spec_run_blocks(c, len - c.id_len - 1u, b);
            if (!n || n > left) return 3;
        }
        b += n;
        if (b >= bend) {
            end = (uint32_t)w.cpos[idx] + len;
            return 2;
        }
        const uint32_t nx = w.csucc[idx];
        if (nx == kS2NoSucc) return (uint32_t)w.cpos[idx] + len >= w.s.limit ? 0u : 4u;
        idx = nx;
        return 1;
    }
    if (b >= bend) {
        end = w.cpos[idx];
        return 2;
    }
    return 1;
}

// Length in bits of the blocks [b0, bend) of an RSI coded from p on (the first CDS carries the
// reference sample when b0 == 0 and the preprocessor is on); 0 = unresolved inside the window.
// Two loops, because on a wavefront a loop costs what its most expensive path costs: first the on-demand
// parses that bring the walk back onto the marked chain (the reference sample of the hypothetical RSI
// start shifted it off), then table steps only -- the successor of a marked boundary is marked, so the
// walk never leaves the chain again.
AEC_HD uint32_t s2_unit(const S2Win &w, const Cfg &c, uint32_t p, uint32_t b0, uint32_t bend, uint32_t budget)
{
    uint32_t pos = p, b = b0;
    uint32_t n_parse = 0, n_table = 0;
    (void)n_parse;
    (void)n_table;
    S2_COUNT(4);
    if (b0 == 0) {
        if (!spec_walk_init(c, spec_first_entry(w.s, c, p), p, pos, b)) return 0;
        if (b > bend) return 0;
    }
    while (b < bend) {                                   // catch-up: on demand until a marked boundary
        if (pos >= w.s.limit) return 0;
        if (s2_marked(w.marks, pos)) break;
        if (budget == 0) { S2_COUNT(5); return 0; }
        budget--;
        n_parse++;
        S2_COUNT(3);
        S2_NOTE(pos);
        const uint32_t e1 = spec_nxt_entry(w.s, c, pos);
        if (!e1) return 0;
        const uint32_t len = e1 & 0xFFFu;
        uint32_t n = 1;
        if (e1 & kNxtZero) {
            n = spec_run_blocks(c, len - c.id_len - 1u, b);
            if (!n || n > bend - b) return 0;
        }
        pos += len;
        b += n;
    }
    if (b < bend) {                                      // (no on-demand parse from here on)
        uint32_t idx = s2_index(w, pos), end = 0;
        if (idx == kS2NoIndex) return 0;
        for (;;) {
            const uint32_t st = s2_table_step(w, c, idx, b, bend, end);
            if (st != 1u && st != 2u) return 0;
            n_table++;
            if (st == 2u) break;
        }
        pos = end;
    }
    S2_HIST(n_parse, n_table);
    return pos - p;
}

// Global record of a candidate (what the walkers read).  Units: short RSIs -- a = the whole RSI;
// long RSIs -- a = the first segment (with the reference sample), m = an inner segment of 64 blocks.
// x / mx: chain of a- / m-units from here until it leaves the window core: bits [0,24) = distance,
// bits [24,32) = units covered (0 = no chain).
struct S2Rec {
    uint32_t a, x, m, mx;
};

}  // namespace aec
