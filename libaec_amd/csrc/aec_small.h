// aec_small.h -- the per-bit arithmetic of the every-bit index scheme (aec_idx.hip: launch_index_small; DESIGN.md §2
// "small streams").  The kernels are loops of these over the bits of a piece of the stream; tests/emul/small_emul.cpp
// runs the same functions bit by bit on the CPU against the RSI starts the oracle's encoder reports.
#pragma once

#include "aec_trunk.h"

namespace aec {

constexpr uint32_t kSmNone = 0xFFFFFFFFu;
constexpr uint32_t kSmHopCds = 8;           // coded data sets per hop

// Step 1: the coded data set that would begin at absolute bit `at`, without (a) and with (b) a reference sample, in the
// nxt[] format of aec_spec.h (length | kNxtBlock or kNxtZero; 0: none that an index pass may follow).  A second-extension
// code beyond the table is a data error to the reference (decode.c:589-616) and to the serial walker (skip_cds): such a
// coded data set does not parse here either -- the chain ends at its RSI and the walker behind gives the verdict.
AEC_HD void sm_parse(const TrStream &s, const Cfg &c, uint64_t at, uint16_t &a, uint16_t &b)
{
    a = b = 0;
    uint32_t nz;
    TrWin W;
    tr_win_load(s, at, W);
    const uint32_t il = c.id_len;
    const uint32_t head = (uint32_t)(tr_peek64(s, at) >> (63u - il));
    const bool se = (head >> 1) == 0u && (head & 1u);
    auto se_ok = [&](uint32_t ref) {
        BitReaderT<MemFetch> br;
        br.init(MemFetch{s.words, s.nwords}, s.end_bit, at + il + 1u + ref * c.bps);
        for (uint32_t k = 0; k < c.bs / 2u; k++) {
            uint32_t m;
            if (!br.unary(m) || m > 90u) return false;
        }
        return true;
    };
    uint32_t len = tr_cds(s, c, at, 0u, nz, W);
    if (len && len < 4096u && (!se || se_ok(0u))) a = (uint16_t)(len | (nz ? kNxtZero : kNxtBlock));
    b = a;
    if (c.flags & F_PREPROCESS) {
        len = tr_cds(s, c, at, 1u, nz, W);
        b = (len && len < 4096u && (!se || se_ok(1u))) ? (uint16_t)(len | (nz ? kNxtZero : kNxtBlock)) : (uint16_t)0;
    }
}

// Hops: from bit q of the piece up to kSmHopCds coded data sets without reference samples -- bits (15) and blocks (9
// bits above) covered.  A run of zero blocks to the end of its segment (run code 5: its length depends on where in the
// RSI it stands, reference decode.c:528-530) ends a hop in front of it.
template <class E0>
AEC_HD uint32_t sm_hop(const Cfg &c, const E0 &e0, uint32_t q, uint32_t nbits)
{
    uint32_t pos = q, blocks = 0;
    for (uint32_t i = 0; i < kSmHopCds && pos < nbits; i++) {
        const uint32_t e = e0(pos);
        const uint32_t len = e & 0xFFFu;
        if (!e || pos + len > nbits) break;
        uint32_t nb = 1;
        if (e & kNxtZero) {
            const uint32_t nz = len - c.id_len - 1u;
            if (nz == 5u) break;
            nb = nz > 5u ? nz - 1u : nz;
        }
        if (blocks + nb > 511u || pos + len - q > 32767u) break;
        pos += len;
        blocks += nb;
    }
    return (pos - q) | (blocks << 15);
}

// Hops of hops, for RSIs of hundreds of blocks: up to 8 hops from bit q -- bits (18) and blocks (14 bits above).
template <class H>
AEC_HD uint32_t sm_hop2(const H &hop, uint32_t q, uint32_t nbits)
{
    uint32_t pos = q, blocks = 0;
    for (uint32_t i = 0; i < 8u && pos < nbits; i++) {
        const uint32_t h = hop(pos);
        const uint32_t hb = h >> 15, hl = h & 0x7FFFu;
        if (!hb || blocks + hb > 16383u || pos + hl - q > 262143u) break;
        pos += hl;
        blocks += hb;
    }
    return (pos - q) | (blocks << 18);
}

// Step 2: one whole RSI from bit q of the piece (its first coded data set with a reference sample, the RSI's own
// bookkeeping of zero-block runs, reference decode.c:518-544): where the next RSI would start, or kSmNone.
// e0 / e1: the parses of step 1; hop, hop2: the hops and the hops of hops (has, has2 == false: none).
// b0: blocks of the RSI that lie in front of q (a walk that resumes inside an RSI; 0: q begins one).
// pad: AEC_PAD_RSI -- the next RSI begins on a byte (decode.c:407-408): 1 + (the piece's first bit modulo 8), 0: no padding.
template <class E0, class E1, class H, class H2>
AEC_HD uint32_t sm_rsi(const Cfg &c, const E0 &e0, const E1 &e1, const H &hop, bool has, const H2 &hop2, bool has2, uint32_t q,
                       uint32_t nbits, uint32_t b0 = 0, uint32_t pad = 0)
{
    const uint32_t rfb = (c.flags & F_PREPROCESS) ? c.bps : 0u;
    uint32_t pos = q, b = b0;
    bool ok = q < nbits;
    for (uint32_t i = 0; ok && i <= c.rsi && b < c.rsi; i++) {
        if (has2 && b != 0u) {
            const uint32_t h = hop2(pos);
            const uint32_t hb = h >> 18;
            if (hb && b + hb <= c.rsi) {
                pos += h & 0x3FFFFu;
                b += hb;
                continue;
            }
        }
        if (has && b != 0u) {
            const uint32_t h = hop(pos);
            const uint32_t hb = h >> 15;
            if (hb && b + hb <= c.rsi) {
                pos += h & 0x7FFFu;
                b += hb;
                continue;
            }
        }
        const uint32_t e = b == 0u ? e1(pos) : e0(pos);
        const uint32_t len = e & 0xFFFu;
        const uint32_t nz = (e & kNxtZero) ? len - c.id_len - 1u - (b == 0u ? rfb : 0u) : 0u;
        const uint32_t nb = e ? tr_blocks(c, nz, b) : 0u;
        ok = nb != 0u && pos + len <= nbits;
        pos += len;
        b += nb;
    }
    if (ok && b == c.rsi && pad) {
        pos = ((pos + pad - 1u + 7u) & ~7u) - (pad - 1u);
        ok = pos <= nbits;
    }
    return (ok && b == c.rsi) ? pos : kSmNone;
}

// Step 3, round k with quarter = 4^k known RSI starts: the starts m * quarter + i, m = 1 .. 3, from start i ...
AEC_HD void sm_double_starts(const uint32_t *j, uint32_t *sidx, uint32_t i, uint32_t quarter, uint32_t scap)
{
    uint32_t a = sidx[i];
    for (uint32_t m = 1; m < 4u; m++) {
        a = a == kSmNone ? kSmNone : j[a];
        if (m * quarter + i < scap) sidx[m * quarter + i] = a;
    }
}
// ... and the table of the next round, j^4
AEC_HD uint32_t sm_double_table(const uint32_t *j, uint32_t q)
{
    uint32_t v = j[q];
    for (uint32_t m = 1; m < 4u && v != kSmNone; m++) v = j[v];
    return v;
}

}  // namespace aec
