// aec_trunk.h -- TRUNK index of a bare stream: per-lane arithmetic (aec_idx.hip: k_trunk, k_trunk_scan,
// k_hyp_walk, k_hyp_land, k_hyp_chain).
//
// A bare CCSDS 121.0-B-2 stream has no entry points: a coded data set (CDS) is found by parsing its
// predecessor (reference src/decode.c:402-421), and where an RSI starts decides how its first CDS is
// parsed (reference sample, decode.c:407-411, 462-502) and how long a rest-of-segment zero run is
// (decode.c:528-530).  The serial walk is replaced by three lane-parallel passes over the stream:
//
//   1. TRUNK.  The CDS parse WITHOUT a reference sample ("N-step") self-synchronises like any prefix
//      code: started at an arbitrary bit it falls onto the true chain of CDS boundaries after a number
//      of codes of the order of the CDS length in bits.  One lane per window of the stream starts
//      `lead` bits in front of its window, burns in, and then records every boundary it visits inside
//      the window: the NODES of the trunk, each with the blocks its CDS covers.  A window whose chain
//      does not begin where the chain of the window before it ended is a SEAM.  A scan over the
//      windows numbers the blocks along the trunk (G), so "the node n blocks behind node x" is a
//      search for G(x) + n -- exact inside a seamless stretch, because there the trunk is ONE path
//      closed under the N-step.
//   2. HYPOTHESES.  The trunk is wrong behind every RSI start (it parsed the CDS that carries the
//      reference sample without one) until it has synchronised again, and it does not know where the
//      RSIs start.  So every node is tried as an RSI start: parse the first CDS with the reference
//      sample, parse on (on demand, with the exact block count, rest-of-segment runs included) until
//      the walk stands on a node of the trunk, then jump the remaining blocks by G.  A walk that
//      completes its RSI before it meets the trunk goes straight on with the next RSI (up to kTrMaxK),
//      so that every record ends on a node: the walker never needs a position that is not one, and the
//      trunk only has to be met once in a while -- it need not be the true chain (where the coded data
//      sets are long it rarely is).  The RSI starts inside such a record are parsed again, by the lane
//      that expands it, only for the records the true walk really takes.
//   3. CHAINS of records inside a window, then the existing wide / serial walkers over windows.
//
// Exactness: every record is the exact value of the reference's walk from its node (the trunk only
// selects WHERE records exist and serves jumps it is closed under); the walkers read records at true
// RSI starts only and fall back to the serial CDS walk where one is missing.  Everything here is
// __host__ __device__: tests/emul runs it lane by lane on the CPU against the oracle.
#pragma once

#include "aec_spec.h"

#ifndef TR_DBG
#define TR_DBG(...) ((void)0)
#endif
#ifndef TR_COUNT_STEP
#define TR_COUNT_STEP(i) ((void)0)      // (emulator statistics)
#endif

namespace aec {

constexpr uint32_t kTrMaxK = 63;              // RSIs one record may cover
constexpr uint32_t kTrMaxScan = 8192;         // longest unary region a table parse follows (bits)
constexpr uint64_t kTrNone = ~0ull;

// per-node word: bits [0,30) = blocks of the window's nodes in front of this one, [30,32) = kind
constexpr uint32_t kTrBpMask = 0x3FFFFFFFu;
constexpr uint32_t kTrRos = 1u;               // rest-of-segment zero run: covers what its RSI position says
constexpr uint32_t kTrDead = 2u;              // no CDS ends inside the stream from here

struct TrStream {
    const uint32_t *words;     // big-endian 32-bit words, 4-byte aligned
    uint64_t nwords;           // readable words (>= 1)
    uint64_t end_bit;          // bits that belong to the stream
};

struct TrGeom {
    uint64_t lo;               // bit position of window 0 (multiple of L)
    uint64_t start_bit;        // where the stream (or the resumed walk) starts: a true CDS boundary
    uint32_t L;                // bits per window, multiple of 32, <= 65536
    uint32_t lead;             // burn-in in front of a region (bits)
    uint32_t ncap;             // node records in all (the windows' nodes are stored back to back)
    uint32_t rw;               // windows per region (one trunk lane)
    uint32_t nwin;             // windows with a trunk (core + look-ahead)
    uint32_t ncore;            // windows whose nodes get records
    uint32_t budget;           // coded data sets one hypothesis may parse
    uint32_t kmax;             // RSIs one record may cover (<= kTrMaxK)
    uint32_t pcap;             // entries of the pool of RSI ends inside records
    uint32_t pad;
};

// Record of a node, the layout the walkers read.
//   x: bits [0,26) = distance to the node the record ends on, [26,32) = RSIs covered; 0 = none
//   y: records of several RSIs: index + 1 of the pool entry that holds the end of the LAST BUT ONE of them
//      (each entry links to the one before: the walkers need the starts inside such a record); 0 = none.
//      Records of ONE RSI: record index + 1 of the node the record ends on (the walker's next record), 0 = unknown
// Between the walk and the jump a hypothesis is parked: x = distance to the node the walk stands on, and
// park[node] = kTrParked | RSIs completed << 16 | blocks of the current RSI in front of that node.
struct TrRec {
    uint32_t x, y;
};
constexpr uint32_t kTrParked = 0x80000000u;

struct TrPoolEntry {           // where an RSI inside a record ends (= where the next one starts, before AEC_PAD_RSI)
    uint32_t end;              // distance from the record's node
    uint32_t prev;             // index + 1 of the entry of the RSI in front, 0 = that was the record's first
};

struct TrTables {
    // trunk (k_trunk)
    uint32_t *bitmap;          // [nwin * L/32]  bit (31 - i % 32) of word i / 32 <=> lo + i is a node
    uint16_t *pre;             // [nwin * L/32]  nodes of ITS window in front of the word
    uint32_t *nbase;           // [nwin + 1]     nodes in front of the window (scan); cpos / bp / rec index
    uint16_t *cpos;            // [ncap]         position inside the window
    uint32_t *bp;              // [ncap]         block prefix | kind << 30
    uint32_t *ccnt;            // [nwin]         nodes of the window (0 after the scan if they do not fit ncap)
    uint32_t *nblk;            // [nwin]         blocks of the window's nodes
    uint32_t *nros;            // [nwin]         rest-of-segment nodes
    uint64_t *entry;           // [nwin]         first boundary of the chain at or behind the window start
    uint64_t *exit;            // [nwin]         first boundary at or behind the window end (kTrNone: the chain died)
    // scan (k_trunk_scan), nwin + 1 entries each
    uint64_t *gbase;           // blocks of the trunk in front of the window
    uint32_t *seampre;         // seams at or in front of the window
    uint32_t *rospre;          // rest-of-segment nodes in front of the window
    // hypotheses
    TrRec *rec;                // [ncap]
    uint32_t *park;            // [ncap]
    TrPoolEntry *pool;         // [pcap]
    uint32_t *pool_cnt;        // entries handed out (may run beyond pcap: then records were dropped)
    const uint32_t *skip_if;   // (device kernels) non-null and *skip_if != 0: another scheme has delivered the index, return
};

AEC_HD uint32_t tr_word(const TrStream &s, uint64_t i)
{
    // (words past the end repeat the last one: a CDS that would need them ends behind end_bit and is rejected)
    return bswap32(s.words[i < s.nwords ? i : s.nwords - 1]);
}

AEC_HD uint64_t tr_peek64(const TrStream &s, uint64_t q)
{
    const uint64_t w = q >> 5;
    const uint32_t sh = (uint32_t)(q & 31u);
    const uint64_t a = ((uint64_t)tr_word(s, w) << 32) | tr_word(s, w + 1);
    const uint64_t lo = sh ? ((uint64_t)tr_word(s, w + 2) >> (32u - sh)) : 0u;
    return (a << sh) | lo;
}

AEC_HD uint32_t tr_popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}

// 256 stream bits from the word that holds bit q on, loaded with two 16-byte requests, so that a coded data
// set of the usual size costs ONE memory round trip (header and unary part come out of registers).
struct TrWin {
    uint64_t q0, q1, q2, q3;      // (named, not an array: indexing one by a lane value would put it in scratch)
};

AEC_HD void tr_win_load(const TrStream &s, uint64_t pos, TrWin &W)
{
    const uint64_t w = pos >> 5;
    uint32_t x[8];
#if defined(__HIP_DEVICE_COMPILE__)
    struct __attribute__((packed, aligned(4))) Q4 {
        uint32_t a, b, c, d;
    };
    if (w + 8u <= s.nwords) {
        const Q4 lo = *reinterpret_cast<const Q4 *>(s.words + w);
        const Q4 hi = *reinterpret_cast<const Q4 *>(s.words + w + 4u);
        x[0] = lo.a; x[1] = lo.b; x[2] = lo.c; x[3] = lo.d;
        x[4] = hi.a; x[5] = hi.b; x[6] = hi.c; x[7] = hi.d;
    } else
#endif
    {
        for (uint32_t i = 0; i < 8u; i++) x[i] = s.words[w + i < s.nwords ? w + i : s.nwords - 1u];
    }
    W.q0 = ((uint64_t)bswap32(x[0]) << 32) | bswap32(x[1]);
    W.q1 = ((uint64_t)bswap32(x[2]) << 32) | bswap32(x[3]);
    W.q2 = ((uint64_t)bswap32(x[4]) << 32) | bswap32(x[5]);
    W.q3 = ((uint64_t)bswap32(x[6]) << 32) | bswap32(x[7]);
}

// 64 bits at bit offset o < 128 of the 192 bits a : b : c.  (Values, not a window indexed by o: a conditional
// between the window's fields selects an ADDRESS and puts the window in scratch.)
AEC_HD uint64_t tr_peek3(uint64_t a, uint64_t b, uint64_t c, uint32_t o)
{
    const bool up = o >= 64u;
    const uint32_t sh = o & 63u;
    const uint64_t hi = up ? b : a, lo = up ? c : b;
    return sh ? (hi << sh) | (lo >> (64u - sh)) : hi;
}

// Length in bits of the CDS that starts at q (0 = none ends inside the stream, or its unary part is
// longer than kTrMaxScan).  `nz` = 0 for a CDS of one block, else the zero-block run code fs + 1
// (reference decode.c:518-536).  Layouts: decode.c:462-502 (split), 589-644 (low entropy), 659-677
// (uncompressed); restated in SURVEY.md Appendix A.  W = tr_win_load(s, q).
// from_memory = false: a unary part that goes on behind the window is not followed (returns 0, *more = true).
AEC_HD uint32_t tr_cds(const TrStream &s, const Cfg &c, uint64_t q, uint32_t ref, uint32_t &nz, const TrWin &W,
                       bool from_memory = true, bool *more = nullptr)
{
    nz = 0;
    if (q + c.id_len >= s.end_bit) return 0;
    const uint32_t sh = (uint32_t)(q & 31u);
    const uint64_t q0 = W.q0, q1 = W.q1, q2 = W.q2, q3 = W.q3;
    const uint64_t H = sh ? (q0 << sh) | (q1 >> (64u - sh)) : q0;
    const uint32_t id = (uint32_t)(H >> (64u - c.id_len));
    if (id == (1u << c.id_len) - 1u) {
        const uint32_t len = c.id_len + c.bs * c.bps;
        return q + len <= s.end_bit ? len : 0u;
    }
    const bool low = id == 0u;
    const uint32_t selbit = (uint32_t)(H >> (63u - c.id_len)) & 1u;
    const uint32_t hdr = c.id_len + (low ? 1u : 0u) + ref * c.bps;
    uint32_t need = low ? (selbit ? c.bs / 2u : 1u) : c.bs - ref;
    const uint32_t add = low ? 0u : need * (id - 1u);
    // unary part: the need-th 1-bit from q + hdr on; 64 bits at a time, out of the window while it lasts
    // Up to three 64-bit pieces of the unary part come out of the window (sh + hdr < 70: the first starts in q0
    // or q1; the third fits if sh + hdr <= 64).  Their popcounts say which piece holds the need-th 1-bit, so
    // ONE select does the work whatever the piece.
    const uint32_t o = sh + hdr;
    const uint64_t U0 = tr_peek3(q0, q1, q2, o), U1 = tr_peek3(q1, q2, q3, o);
    const uint64_t U2 = o <= 64u ? tr_peek3(q2, q3, 0u, o) : 0u;
    const uint32_t p0 = tr_popc64(U0), p1 = tr_popc64(U1), p2 = tr_popc64(U2);
    const bool in0 = p0 >= need, in1 = p0 + p1 >= need, in2 = o <= 64u && p0 + p1 + p2 >= need;
    uint32_t used = hdr;                                 // bits of the CDS in front of the unary bits still to look at
    bool found = in0 || in1 || in2;
    if (found) {
        const uint64_t U = in0 ? U0 : (in1 ? U1 : U2);
        const uint32_t skip = in0 ? 0u : (in1 ? p0 : p0 + p1);
        used += (in0 ? 0u : (in1 ? 64u : 128u)) + spec_select64(U, need - skip) + 1u;
    } else {
        const uint32_t pieces = o <= 64u ? 3u : 2u;
        need -= p0 + p1 + (o <= 64u ? p2 : 0u);
        used += 64u * pieces;
    }
    if (!found && !from_memory) {
        if (more) *more = true;
        return 0;
    }
    while (!found) {                                     // long unary parts: from memory
        const uint64_t p = q + used;
        if (p >= s.end_bit || used > kTrMaxScan) return 0;
        const uint64_t U = tr_peek64(s, p);
        const uint32_t pc = tr_popc64(U);
        if (pc >= need) {
            used += spec_select64(U, need) + 1u;
            found = true;
        } else {
            need -= pc;
            used += 64u;
        }
    }
    if (low && !selbit) nz = used - hdr;
    const uint64_t len = (uint64_t)used + add;
    return q + len <= s.end_bit ? (uint32_t)len : 0u;
}

AEC_HD uint32_t tr_cds(const TrStream &s, const Cfg &c, uint64_t q, uint32_t ref, uint32_t &nz)
{
    TrWin W;
    tr_win_load(s, q, W);
    return tr_cds(s, c, q, ref, nz, W);
}

// blocks covered by a CDS with run code nz (0: not a zero-block CDS) at block b of its RSI; 0 = the
// run overruns the RSI (AEC_DATA_ERROR in the reference, decode.c:543-544)
AEC_HD uint32_t tr_blocks(const Cfg &c, uint32_t nz, uint32_t b)
{
    if (nz == 0u) return 1u;
    return spec_run_blocks(c, nz, b);
}

// ---- 1. trunk ----------------------------------------------------------------------------------------
// The chain through window w, entering at `pos` (its first boundary at or behind the window start; kTrNone:
// no chain); returns where the chain leaves the window.  The nodes are stored back to back, so a window is
// walked twice: COUNT writes bitmap, prefix counts and the window's totals (then the scan places the windows'
// nodes), FILL writes position and block prefix of every node.
enum : uint32_t { TR_COUNT = 0, TR_FILL = 1 };

AEC_HD uint64_t tr_trunk_window(const TrStream &s, const Cfg &c, const TrGeom &g, const TrTables &t, uint32_t w,
                                uint64_t pos, uint64_t *exit_out, uint32_t mode)
{
    const uint64_t wstart = g.lo + (uint64_t)w * g.L, wend = wstart + g.L;
    const uint32_t nw = g.L / 32u;
    uint32_t *bm = t.bitmap + (uint64_t)w * nw;
    uint16_t *pre = t.pre + (uint64_t)w * nw;
    if (mode == TR_FILL) {
        const uint32_t n = t.ccnt[w];
        if (!n) {                            // nothing to store; a window the scan dropped loses its marks as well
            if (pos != kTrNone && pos < wend)
                for (uint32_t i = 0; i < nw; i++) {
                    bm[i] = 0;
                    pre[i] = 0;
                }
            while (pos != kTrNone && pos < wend) {
                uint32_t nzc;
                const uint32_t len = tr_cds(s, c, pos, 0u, nzc);
                pos = len ? pos + len : kTrNone;
            }
            return pos;
        }
        uint16_t *cpos = t.cpos + t.nbase[w];
        uint32_t *bp = t.bp + t.nbase[w];
        uint32_t cnt = 0, blocks = 0;
        while (pos != kTrNone && pos < wend) {
            uint32_t nzc;
            const uint32_t len = tr_cds(s, c, pos, 0u, nzc);
            if (cnt < n) {
                uint32_t nb = 1, kind = 0;
                if (!len) {
                    nb = 0;
                    kind = kTrDead;
                } else if (nzc == 5u) {
                    kind = kTrRos;
                } else if (nzc) {
                    nb = nzc > 5u ? nzc - 1u : nzc;
                }
                cpos[cnt] = (uint16_t)(pos - wstart);
                bp[cnt] = blocks | (kind << 30);
                cnt++;
                blocks += nb;
            }
            pos = len ? pos + len : kTrNone;
        }
        return pos;
    }
    t.entry[w] = pos;
    uint32_t cnt = 0, blocks = 0, ros = 0, wi = 0, wv = 0, wpre = 0;
    bool full = false;               // more blocks than a prefix holds: the rest of the window has no trunk
    while (pos != kTrNone && pos < wend) {
        uint32_t nzc;
        const uint32_t len = tr_cds(s, c, pos, 0u, nzc);
        if (!full && blocks > kTrBpMask - 64u) full = true;
        if (!full) {
            const uint32_t rel = (uint32_t)(pos - wstart), word = rel >> 5;
            while (wi < word) {
                bm[wi] = wv;
                pre[wi] = (uint16_t)wpre;
                wi++;
                wv = 0;
                wpre = cnt;
            }
            uint32_t nb = 1;
            if (!len) nb = 0;                                   // (dead end: still a node, covers nothing)
            else if (nzc == 5u) ros++;                          // (rest of segment: counts one block on the trunk)
            else if (nzc) nb = nzc > 5u ? nzc - 1u : nzc;
            wv |= 0x80000000u >> (rel & 31u);
            cnt++;
            blocks += nb;
        }
        pos = len ? pos + len : kTrNone;
    }
    while (wi < nw) {
        bm[wi] = wv;
        pre[wi] = (uint16_t)wpre;
        wi++;
        wv = 0;
        wpre = cnt;
    }
    exit_out[w] = full ? kTrNone : pos;
    t.ccnt[w] = cnt;
    t.nblk[w] = blocks;
    t.nros[w] = ros;
    return pos;
}

// A run of uncompressed coded data sets right in front of `upto`: the first of eight headers -- all ones -- a fixed
// distance apart, or kTrNone.  Such a run is the true chain wherever it is met (chance aside, and the repair passes
// check every region's entry anyway), and on raw samples without the preprocessor -- every block uncompressed, the
// same bits at the same place in every sample -- it is the ONLY way onto it: a chain that is off there stays off.
AEC_HD uint64_t tr_unc_run(const TrStream &s, const Cfg &c, uint64_t start_bit, uint64_t upto)
{
    const uint32_t step = c.id_len + c.bs * c.bps, idmax = (1u << c.id_len) - 1u;
    const uint64_t span = 9ull * step;
    if (upto < start_bit + span || upto > s.end_bit) return kTrNone;
    const uint64_t from = upto - span;
    auto id_at = [&](uint64_t q) -> uint32_t {
        const uint64_t wi = q >> 5;
        const uint64_t two = ((uint64_t)tr_word(s, wi) << 32) | tr_word(s, wi + 1u);
        return (uint32_t)((two << (q & 31u)) >> (64u - c.id_len));
    };
    for (uint32_t k = 0; k < step; k++) {
        const uint64_t q = from + k;
        if (id_at(q) != idmax) continue;
        bool run = true;
        for (uint32_t j = 1; j < 8u && run; j++) run = id_at(q + (uint64_t)j * step) == idmax;
        if (run) return q;
    }
    return kTrNone;
}

// One trunk lane: region r = windows [r * rw, ...).  exit_prev == nullptr: the first pass (burn-in from `lead`
// bits in front of the region); else a repair pass: a region whose first window does not begin where the
// window in front of it ended (exit_prev, the result of the pass before) is walked again from there.
AEC_HD void tr_trunk_region(const TrStream &s, const Cfg &c, const TrGeom &g, const TrTables &t, uint32_t r,
                            const uint64_t *exit_prev, uint64_t *exit_out, uint32_t mode = TR_COUNT)
{
    const uint32_t w0 = r * g.rw, w1 = w0 + g.rw < g.nwin ? w0 + g.rw : g.nwin;
    const uint64_t rstart = g.lo + (uint64_t)w0 * g.L, rend = g.lo + (uint64_t)w1 * g.L;
    uint64_t pos;
    if (mode == TR_FILL) {
        pos = t.entry[w0];
    } else if (!exit_prev) {
        pos = rstart > g.start_bit + g.lead ? rstart - g.lead : g.start_bit;
        if (rend <= g.start_bit || rstart > s.end_bit) pos = kTrNone;
        if (pos != kTrNone && pos != g.start_bit) {
            const uint64_t q = tr_unc_run(s, c, g.start_bit, rstart);
            if (q != kTrNone) pos = q;                                   // (nine steps of burn-in instead of thousands)
        }
        while (pos != kTrNone && pos < rstart) {                         // burn-in
            uint32_t nzc;
            const uint32_t len = tr_cds(s, c, pos, 0u, nzc);
            pos = len ? pos + len : kTrNone;
        }
    } else {
        const uint64_t prev = w0 ? exit_prev[w0 - 1u] : kTrNone;
        if (prev == kTrNone || prev == t.entry[w0]) {
            for (uint32_t w = w0; w < w1; w++) exit_out[w] = exit_prev[w];
            return;
        }
        pos = prev;
    }
    for (uint32_t w = w0; w < w1; w++) {
        // (a repair that is back on the chain the region held before: the windows from here on stand as they are)
        if (exit_prev && mode == TR_COUNT && w > w0 && pos == t.entry[w]) {
            for (uint32_t u = w; u < w1; u++) exit_out[u] = exit_prev[u];
            return;
        }
        pos = tr_trunk_window(s, c, g, t, w, pos, exit_out, mode);
    }
}

// scan over the windows (serial form; the kernel does the same with a workgroup scan)
AEC_HD void tr_scan_serial(const TrGeom &g, const TrTables &t)
{
    uint64_t gsum = 0, nodes = 0;        // nodes: unclipped, so that the rule does not depend on the order
    uint32_t seams = 0, ros = 0;
    for (uint32_t w = 0; w < g.nwin; w++) {
        // a window whose nodes do not fit the record space has none, and the window behind it is a seam
        const bool fits = nodes + t.ccnt[w] <= g.ncap;
        const bool seam = w == 0 || t.exit[w - 1] == kTrNone || t.exit[w - 1] != t.entry[w] || nodes > g.ncap;
        seams += seam ? 1u : 0u;
        t.nbase[w] = (uint32_t)(nodes < g.ncap ? nodes : g.ncap);
        nodes += t.ccnt[w];
        if (!fits) {
            t.ccnt[w] = 0;
            t.nblk[w] = 0;
            t.nros[w] = 0;
        }
        t.gbase[w] = gsum;
        t.seampre[w] = seams;
        t.rospre[w] = ros;
        gsum += t.nblk[w];
        ros += t.nros[w];
    }
    if (nodes > g.ncap) nodes = g.ncap;
    t.nbase[g.nwin] = (uint32_t)nodes;
    t.gbase[g.nwin] = gsum;
    t.seampre[g.nwin] = seams + 1u;
    t.rospre[g.nwin] = ros;
}

// node index of absolute bit position p inside the tabulated range (false: not a node)
AEC_HD bool tr_node_at(const TrGeom &g, const TrTables &t, uint64_t p, uint32_t &w, uint32_t &idx)
{
    if (p < g.lo) return false;
    const uint64_t i = p - g.lo;
    const uint64_t wi = i / g.L;
    if (wi >= g.nwin) return false;
    const uint32_t word = t.bitmap[i >> 5], sh = (uint32_t)(i & 31u);
    if (!((word >> (31u - sh)) & 1u)) return false;
    w = (uint32_t)wi;
    idx = (uint32_t)t.pre[i >> 5] + (sh ? spec_popc(word >> (32u - sh)) : 0u);
    return true;
}

AEC_HD bool tr_marked(const TrGeom &g, const TrTables &t, uint64_t p)
{
    if (p < g.lo) return false;
    const uint64_t i = p - g.lo;
    if (i >= (uint64_t)g.nwin * g.L) return false;
    return (t.bitmap[i >> 5] >> (31u - (uint32_t)(i & 31u))) & 1u;
}

// ---- 2. hypotheses ------------------------------------------------------------------------------------
// Where a walk gets its stream bits and trunk marks from.  TrGlobal: device memory.  TrStaged: a stretch of
// the stream a workgroup has staged in LDS (plain arrays in the emulator) together with the trunk marks and,
// per bit position, the length of the coded data set that starts there if it is an ordinary one -- one
// block, no reference sample, at most 255 bits -- so that such a step of a walk is ONE byte read; everything
// else (and every position outside the stretch) takes the parse.
struct TrGlobal {
    const TrStream &s;
    const TrGeom &g;
    const TrTables &t;
    AEC_HD bool marked(uint64_t p) const { return tr_marked(g, t, p); }
    AEC_HD void win(uint64_t p, TrWin &W) const { tr_win_load(s, p, W); }
    AEC_HD uint32_t fast(uint64_t) const { return 0u; }
};

struct TrStaged {
    TrGlobal mem;
    const uint32_t *sw;        // stream words in host order, words [0, bits / 32 + 8)
    const uint32_t *bm;        // trunk marks, bits / 32 words
    const uint8_t *nx;         // ordinary coded data set length per bit position, 0 = take the parse
    uint64_t base;             // bit position of sw[0] (multiple of 32)
    uint32_t bits;             // staged bits (multiple of 32)
    AEC_HD bool in(uint64_t p) const { return p >= base && p - base < bits; }
    AEC_HD bool marked(uint64_t p) const
    {
        if (!in(p)) return mem.marked(p);
        const uint32_t r = (uint32_t)(p - base);
        return (bm[r >> 5] >> (31u - (r & 31u))) & 1u;
    }
    AEC_HD void win(uint64_t p, TrWin &W) const
    {
        if (!in(p)) {
            mem.win(p, W);
            return;
        }
        const uint32_t *x = sw + ((uint32_t)(p - base) >> 5);
        W.q0 = ((uint64_t)x[0] << 32) | x[1];
        W.q1 = ((uint64_t)x[2] << 32) | x[3];
        W.q2 = ((uint64_t)x[4] << 32) | x[5];
        W.q3 = ((uint64_t)x[6] << 32) | x[7];
    }
    AEC_HD uint32_t fast(uint64_t p) const
    {
        const uint32_t e = (nx && in(p)) ? nx[(uint32_t)(p - base)] : 0u;
        TR_COUNT_STEP(e ? 0 : (in(p) ? 1 : 2));
        return e;
    }
};

// entry of TrStaged::nx for bit position q (W = the window at q)
AEC_HD uint8_t tr_fast_entry(const TrStream &s, const Cfg &c, uint64_t q, const TrWin &W)
{
    uint32_t nz;
    const uint32_t len = tr_cds(s, c, q, 0u, nz, W, false);
    return (len && !nz && len < 256u) ? (uint8_t)len : (uint8_t)0;
}

struct TrHyp {                 // one hypothesis walk in flight
    uint64_t c;                // the node tried as an RSI start
    uint64_t pos;              // where the next CDS starts
    uint32_t b, k, steps;      // blocks of the current RSI done, RSIs completed, coded data sets parsed
    uint32_t link;             // pool entry (index + 1) of the last RSI end recorded, 0 = none
    uint32_t pend;             // an RSI ended off the trunk at c + pend: the caller records it (tr_hyp_commit)
};

enum : uint32_t { TR_RUN = 0, TR_LAND = 1, TR_DONE = 2, TR_FAIL = 3 };

AEC_HD uint64_t tr_rsi_start(const Cfg &c, uint64_t end_of_previous)     // reference decode.c:407-408
{
    return (c.flags & F_PAD_RSI) ? (end_of_previous + 7u) & ~7ull : end_of_previous;
}

// c: the node tried as the place where an RSI starts (with AEC_PAD_RSI: where the RSI in front of it ends)
AEC_HD void tr_hyp_start(const Cfg &cfg, TrHyp &h, uint64_t c)
{
    h.c = c;
    h.pos = tr_rsi_start(cfg, c);
    h.b = 0;
    h.k = 0;
    h.steps = 0;
    h.link = 0;
    h.pend = 0;
}

// A walk moves on by a coded data set of `len` bits that covers `nb` blocks.  TR_DONE: h.k whole RSIs, ending
// on a node at h.pos.  TR_RUN with h.pend set: an RSI ended off the trunk and the walk goes on -- the caller
// records the end with tr_hyp_commit() before the next step (the kernels allocate the pool entries of a
// wavefront together).
// (tr_hyp_complete: the part that follows when the step completed an RSI -- h.b == c.rsi; `on_trunk` = is
// h.pos a node)
AEC_HD uint32_t tr_hyp_complete(const Cfg &c, const TrGeom &g, TrHyp &h, bool on_trunk)
{
    h.k++;
    if (on_trunk) return TR_DONE;
    if (h.k >= g.kmax || h.pos - h.c > 0xFFFFFFFFull) return TR_FAIL;
    h.pend = (uint32_t)(h.pos - h.c);                     // the walk goes on: where this RSI ended is recorded
    h.b = 0;
    h.pos = tr_rsi_start(c, h.pos);
    return TR_RUN;
}

template <class Src>
AEC_HD uint32_t tr_hyp_advance(const Cfg &c, const TrGeom &g, const Src &src, TrHyp &h, uint32_t len, uint32_t nb)
{
    h.pos += len;
    h.b += nb;
    h.steps++;
    if (h.b == c.rsi) return tr_hyp_complete(c, g, h, src.marked(h.pos));
    return TR_RUN;
}

// the step by the parse: W = the window at h.pos
AEC_HD uint32_t tr_hyp_parse(const TrStream &s, const Cfg &c, const TrHyp &h, const TrWin &W, uint32_t &nb,
                             bool from_memory = true, bool *more = nullptr)
{
    const uint32_t ref = (h.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
    uint32_t nz;
    const uint32_t len = tr_cds(s, c, h.pos, ref, nz, W, from_memory, more);
    if (!len) return 0;
    nb = tr_blocks(c, nz, h.b);
    return (nb && nb <= c.rsi - h.b) ? len : 0u;
}

// One coded data set of the walk.  TR_LAND: the walk stands on a node with blocks of its RSI left (the jump
// takes over); else as tr_hyp_advance.  (The kernels run the pieces of this in separate phases of a wavefront,
// by what a step needs: a byte of the table, the parse on staged words, or device memory.)
template <class Src>
AEC_HD uint32_t tr_hyp_step(const TrStream &s, const Cfg &c, const TrGeom &g, const Src &src, TrHyp &h)
{
    const bool first = h.b == 0u;
    if (!first && src.marked(h.pos)) return TR_LAND;
    if (h.steps >= g.budget) return TR_FAIL;
    uint32_t len = (first && (c.flags & F_PREPROCESS)) ? 0u : src.fast(h.pos), nb = 1;
    if (!len) {
        TrWin W;
        src.win(h.pos, W);
        len = tr_hyp_parse(s, c, h, W, nb);
        if (!len) return TR_FAIL;
    }
    return tr_hyp_advance(c, g, src, h, len, nb);
}

// records the pending RSI end in pool entry e (0xFFFFFFFF: none left -- the hypothesis fails); false = failed
AEC_HD bool tr_hyp_commit(const TrGeom &g, const TrTables &t, TrHyp &h, uint32_t e)
{
    if (e >= g.pcap) return false;
    t.pool[e] = TrPoolEntry{h.pend, h.link};
    h.link = e + 1u;
    h.pend = 0;
    return true;
}

AEC_HD uint32_t tr_rec_pack(uint64_t bits, uint32_t k)
{
    return (bits && bits < (1u << 26) && k >= 1u && k <= kTrMaxK) ? (uint32_t)bits | (k << 26) : 0u;
}
AEC_HD uint32_t tr_rec_bits(uint32_t x) { return x & 0x03FFFFFFu; }
AEC_HD uint32_t tr_rec_k(uint32_t x) { return x >> 26; }

// what the walk of node (w, idx) leaves in its record
AEC_HD void tr_hyp_finish(const TrGeom &g, const TrTables &t, uint32_t w, uint32_t idx, const TrHyp &h, uint32_t state)
{
    TrRec r{0u, 0u};
    uint32_t park = 0;
    const uint64_t d = h.pos - h.c;
    if (state == TR_DONE) {
        r.x = tr_rec_pack(d, h.k);
        // (the last RSI ends on the node itself; if the walk recorded the end of the one in front, the list
        // starts there.  A record of one RSI has no list.)
        r.y = r.x ? h.link : 0u;
    } else if (state == TR_LAND && d <= 0xFFFFFFFFull) {
        r.x = (uint32_t)d;
        r.y = h.link;
        park = kTrParked | (h.k << 16) | h.b;           // (b < rsi <= 4096, k < 64)
    }
    t.rec[t.nbase[w] + idx] = r;
    t.park[t.nbase[w] + idx] = park;
}

// ---- 2b. hypothesis walks that COALESCE ------------------------------------------------------------------
// A walk from node x -- the coded data set with the reference sample, then parses without one -- is a garbage
// parse until it falls onto the trunk, about one coded-data-set length in coded data sets later (253 at 253 bits).
// But the POSITIONS such a walk visits do not depend on how many blocks of its RSI it has behind it (the parse
// without a reference sample never asks; only a rest-of-segment run counts its blocks by it, and its length in
// bits does not), so two walks that meet go the same way from there on: every walk marks the positions it parses
// at, a walk that arrives on a mark stops and takes the owner's result, shifted by the difference of their block
// counts.  A walk then runs until it meets ANY other walk, not the one trunk: a dozen parses instead of a few
// hundred (measured in tests/emul: 12 per node at 253 bits per coded data set, 280 without).  Marks live in the
// LDS of the workgroup that walks a stretch of the stream (k_hyp_walk_co), so owners are nodes of the same
// stretch; a walk that has not met anything after `tmax` parses, or leaves the staged stretch, is handed on to a
// kernel that finishes it from device memory (k_hyp_walk_rest), and whoever met its marks waits for that.
//
// Rest-of-segment runs: a walk stops MARKING at its first one (its block count jumps to the next multiple of 64,
// a different jump for every walk that shares the path).  Who took a mark in front of it makes that jump with its
// own count; from there on both counts are multiples of 64 plus the same number, later runs move them alike, and
// what the owner gained behind its first run is gained by the other as well.
//
// How a walk can end without a landing: CO_FAIL -- a coded data set on its way does not parse (whoever shares the
// way fails with it: no record, which is the plain walk's answer too); CO_OVER -- its block count reaches the
// end of its RSI on the way, or a zero run would overrun it: the plain walk would go on with the next RSI
// (tr_hyp_step; records of several RSIs) -- what the caller does with those is its choice (where RSIs are many
// times longer than the way back to the trunk they are garbage walks whose counts zero-run codes have inflated:
// a true RSI start never takes that long); CO_PLAIN -- anything this scheme has no room for.  Exactness never
// depends on any of this: a node without a record is walked serially if it is a true RSI start.
constexpr uint32_t kCoTag = 0x40000000u;      // park word of a node: a coalescing record, not a parked hypothesis
constexpr uint32_t kCoNoRos = 0x1FFFu;
enum : uint32_t { CO_RUN = 0, CO_LAND = 1, CO_LINK = 2, CO_QUEUE = 3, CO_FAIL = 4, CO_PLAIN = 5, CO_DEFER = 6, CO_OVER = 7 };
// (tr_co_resolve only, never in a record: a guest of a walk that ran over, with room left in its own RSI, goes on
// from where that walk stopped -- handed on like a root)
constexpr uint32_t CO_GOON = 8;

AEC_HD uint32_t co_pack(uint32_t kind, uint32_t b_end, uint32_t b_ros)
{
    return kCoTag | (kind << 26) | ((b_ros & 0x1FFFu) << 13) | (b_end & 0x1FFFu);
}
AEC_HD uint32_t co_kind(uint32_t p) { return (p >> 26) & 7u; }
AEC_HD uint32_t co_bend(uint32_t p) { return p & 0x1FFFu; }
AEC_HD uint32_t co_bros(uint32_t p) { return (p >> 13) & 0x1FFFu; }

// a mark: owner + 1 in bits [0,11), the owner's block count there in [11,24), position inside the cell in [24,28)
constexpr uint32_t kCoMaxOwners = 2047;
AEC_HD uint32_t co_mark(uint32_t owner, uint32_t b, uint32_t sub) { return (owner + 1u) | (b << 11) | (sub << 24); }
AEC_HD uint32_t co_mark_owner(uint32_t m) { return (m & 0x7FFu) - 1u; }
AEC_HD uint32_t co_mark_b(uint32_t m) { return (m >> 11) & 0x1FFFu; }
AEC_HD uint32_t co_mark_sub(uint32_t m) { return (m >> 24) & 0xFu; }

struct CoWalk {
    uint64_t c, pos;           // the node, where the next coded data set starts
    uint32_t b, steps, b_ros;  // blocks of the RSI done, parses, block count in front of the first rest-of-segment run
};

AEC_HD void tr_co_start(const Cfg &cfg, CoWalk &h, uint64_t c)
{
    h.c = c;
    h.pos = tr_rsi_start(cfg, c);
    h.b = 0;
    h.steps = 0;
    h.b_ros = kCoNoRos;
}

// block count behind a rest-of-segment run that starts at block b (reference decode.c:528-530)
AEC_HD uint32_t co_after_ros(const Cfg &c, uint32_t b)
{
    const uint32_t e = (b & ~63u) + 64u;
    return e < c.rsi ? e : c.rsi;
}

// One step of a coalescing walk.  Src: bits and trunk marks (TrStaged over the stretch).  Cells: the marks of
// the stretch -- claim(cell, payload) = what the cell held before (0: it is ours now), peek(cell).  `lim` = first
// position the stretch does not serve (marks, bits with their look-ahead).  CO_LINK: `hit` = the mark met.
template <class Src, class Cells>
AEC_HD uint32_t tr_co_step(const TrStream &s, const Cfg &c, const Src &src, Cells &cells, CoWalk &h, uint32_t owner,
                           uint64_t base, uint64_t lim, uint32_t shift, uint32_t tmax, uint32_t &hit)
{
    if (h.pos >= lim) return h.b ? CO_QUEUE : CO_PLAIN;     // (a node whose first coded data set lies outside: plain walk)
    if (h.b != 0u) {
        if (src.marked(h.pos)) return CO_LAND;
        const uint32_t rel = (uint32_t)(h.pos - base), cell = rel >> shift, sub = rel & ((1u << shift) - 1u);
        const uint32_t old = h.b_ros == kCoNoRos ? cells.claim(cell, co_mark(owner, h.b, sub)) : cells.peek(cell);
        if (old && co_mark_sub(old) == sub && co_mark_owner(old) != owner) {
            hit = old;
            return CO_LINK;
        }
        if (h.steps >= tmax) return CO_QUEUE;
    }
    const uint32_t ref = (h.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
    TrWin W;
    src.win(h.pos, W);
    uint32_t nz;
    const uint32_t len = tr_cds(s, c, h.pos, ref, nz, W);
    if (!len) return CO_FAIL;
    const uint32_t nb = tr_blocks(c, nz, h.b);
    if (!nb || nb > c.rsi - h.b) return CO_OVER;
    if (nz == 5u && h.b_ros == kCoNoRos) h.b_ros = h.b;
    h.pos += len;
    h.b += nb;
    h.steps++;
    return h.b == c.rsi ? CO_OVER : CO_RUN;
}

// What a walk leaves (workgroup-local form): k = co_pack(kind, block count at its end, b_ros);
// t = CO_LAND / CO_QUEUE: distance from its node to where it stands; CO_LINK: the mark it met.
struct CoRec {
    uint32_t k, t;
};

// A guest with block count b at a mark the owner made with count bm rides to the owner's end: its count there
// (0 = it would complete or overrun its RSI on the way).
AEC_HD uint32_t co_ride(const Cfg &c, uint32_t b, uint32_t bm, uint32_t owner_k)
{
    const uint32_t be = co_bend(owner_k), br = co_bros(owner_k);
    uint32_t out;
    if (br == kCoNoRos || br < bm) {
        out = b + (be - bm);
    } else {
        const uint32_t mine = co_after_ros(c, b + (br - bm)), theirs = co_after_ros(c, br);
        if (mine >= c.rsi) return 0u;
        out = mine + (be - theirs);
    }
    return out < c.rsi ? out : 0u;
}

// Follows the links of walk n inside the workgroup.  CO_LAND: z = where it lands (absolute), b = its count there;
// CO_DEFER: root = the walk (local number) it ends up waiting for, b = its count where that one was handed on;
// CO_QUEUE: n itself is handed on; else CO_PLAIN.
// CO_GOON: the way n took ends where another walk ran over the end of ITS RSI; n has b blocks there (fewer) and
// goes on from z by itself.
template <class Recs, class NodePos>
AEC_HD uint32_t tr_co_resolve(const Cfg &c, Recs recs, NodePos node_pos, uint32_t n, uint64_t &z, uint32_t &root,
                              uint32_t &b)
{
    CoRec r = recs(n);
    uint32_t kind = co_kind(r.k), cur = n;
    b = co_bend(r.k);
    for (uint32_t hops = 0; hops < 64u; hops++) {
        if (kind == CO_LAND) {
            z = node_pos(cur) + r.t;
            return CO_LAND;
        }
        if (kind == CO_QUEUE) {
            root = cur;
            return cur == n ? CO_QUEUE : CO_DEFER;
        }
        if (kind == CO_FAIL) return CO_FAIL;
        if (kind == CO_OVER) {
            if (cur == n) return CO_OVER;
            z = node_pos(cur) + r.t;                        // (a guest with room left: on from where the owner stopped)
            return CO_GOON;
        }
        if (kind != CO_LINK) return CO_PLAIN;
        const uint32_t q = co_mark_owner(r.t), bm = co_mark_b(r.t);
        const CoRec o = recs(q);
        b = co_ride(c, b, bm, o.k);
        if (!b) return CO_OVER;
        // (a way that ends in a coded data set that does not parse fails whoever reaches that point inside its RSI)
        if (co_kind(o.k) == CO_FAIL) return CO_FAIL;
        cur = q;
        r = o;
        kind = co_kind(r.k);
    }
    return CO_PLAIN;
}

// A walk that was handed on (k_hyp_walk_rest): from (pos, b) until it stands on the trunk.  Returns the record's
// park word; dist = where it stands, from the node.
template <class Src>
AEC_HD uint32_t tr_co_rest(const TrStream &s, const Cfg &c, const TrGeom &g, const Src &src, uint64_t node, uint64_t pos,
                           uint32_t b, uint32_t b_ros, uint32_t &dist, uint32_t *parses = nullptr)
{
    const uint32_t b0 = b;
    for (uint32_t steps = 0; steps < g.budget; steps++) {
        if (parses) *parses = steps;
        if (src.marked(pos)) {
            if (pos - node > 0xFFFFFFFFull) return co_pack(CO_PLAIN, 0u, kCoNoRos);
            dist = (uint32_t)(pos - node);
            return co_pack(CO_LAND, b, b_ros);
        }
        TrWin W;
        src.win(pos, W);
        uint32_t nz;
        const uint32_t len = tr_cds(s, c, pos, 0u, nz, W);
        if (!len) return co_pack(CO_FAIL, b, b_ros);
        const uint32_t nb = tr_blocks(c, nz, b);
        // (a walk that runs over the end of its RSI: b_end = 1 if its block count grew about as its coded data sets
        // did -- at most half as many blocks again: no zero-run codes inflating it, as on a TRUE chain that the trunk
        // has not found again for a whole RSI (one RSI in a few thousand at 253 bits per coded data set) -- and 0 for
        // the garbage walks, which reach the end of their RSI in a quarter of its blocks' worth of coded data sets)
        if (!nb || nb > c.rsi - b) return co_pack(CO_OVER, (steps + 1u) * 3u >= (b - b0) * 2u ? 1u : 0u, kCoNoRos);
        if (nz == 5u && b_ros == kCoNoRos) b_ros = b;
        pos += len;
        b += nb;
        if (b == c.rsi) return co_pack(CO_OVER, (steps + 1u) * 3u >= (b - b0) * 2u ? 1u : 0u, kCoNoRos);
    }
    return co_pack(CO_PLAIN, 0u, kCoNoRos);
}

// A guest that waited for a handed-on walk: its count was b where that walk was handed on with count bh; the
// walk's final record is root_k.  Its count where the walk landed (0 = not to be had).
// kind: CO_LAND (with the count), or how it ends instead (CO_FAIL / CO_OVER / CO_PLAIN, as tr_co_resolve; a guest
// with a lower count than a root that ran over is CO_PLAIN: it may still land)
AEC_HD uint32_t co_defer(const Cfg &c, uint32_t b, uint32_t bh, uint32_t root_k, uint32_t &kind)
{
    const uint32_t rk = co_kind(root_k);
    // (a root that ran over as a true chain would: its guests get the plain walk, whatever their counts)
    kind = rk == CO_OVER ? ((b >= bh && !co_bend(root_k)) ? CO_OVER : CO_PLAIN) : CO_PLAIN;
    if (rk != CO_LAND && rk != CO_FAIL) return 0u;
    const uint32_t out = co_ride(c, b, bh, root_k);
    kind = !out ? CO_OVER : (rk == CO_FAIL ? CO_FAIL : CO_LAND);
    return rk == CO_LAND ? out : 0u;
}

// window that holds block number G of the trunk, searched from window w on (g.nwin: none)
AEC_HD uint32_t tr_window_of(const TrGeom &g, const TrTables &t, uint32_t w, uint64_t G)
{
    // gallop, then bisect: gbase is non-decreasing, the window is the LAST one with gbase <= G (an empty
    // window shares its gbase with its successor)
    uint32_t lo = w, step = 1;
    uint32_t hi = w + 1;
    while (hi < g.nwin && t.gbase[hi] <= G) {
        lo = hi;
        hi = hi + step < g.nwin ? hi + step : g.nwin;
        step *= 2;
    }
    // invariant: gbase[lo] <= G, (hi == nwin or gbase[hi] > G)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (t.gbase[mid] <= G) lo = mid;
        else hi = mid;
    }
    return G < t.gbase[lo + 1u] ? lo : g.nwin;
}

// node of window w whose block prefix is exactly `want` (false: none -- the block lies inside a zero run)
AEC_HD bool tr_node_of(const TrGeom &g, const TrTables &t, uint32_t w, uint32_t want, uint32_t &idx)
{
    const uint32_t *bp = t.bp + t.nbase[w];
    const uint32_t n = t.ccnt[w];
    if (!n) return false;
    uint32_t hi = want < n - 1u ? want : n - 1u;            // bp[i] >= i
    if ((bp[hi] & kTrBpMask) == want) {
        idx = hi;
        return true;
    }
    uint32_t lo = 0;
    while (lo < hi) {                                       // first index with bp >= want, in [lo, hi]
        const uint32_t mid = lo + (hi - lo) / 2u;
        if ((bp[mid] & kTrBpMask) < want) lo = mid + 1u;
        else hi = mid;
    }
    if ((bp[lo] & kTrBpMask) != want) return false;
    idx = lo;
    return true;
}

// first rest-of-segment node at or behind node (w, i) (false: none inside the tables)
AEC_HD bool tr_next_ros(const TrGeom &g, const TrTables &t, uint32_t w, uint32_t i, uint32_t &rw, uint32_t &ri)
{
    if (t.rospre[g.nwin] == t.rospre[w]) return false;
    if (t.nros[w]) {
        const uint32_t *bp = t.bp + t.nbase[w];
        const uint32_t n = t.ccnt[w];
        for (uint32_t j = i; j < n; j++)
            if ((bp[j] >> 30) == kTrRos) {
                rw = w;
                ri = j;
                return true;
            }
    }
    // first later window with one: smallest v > w with rospre[v + 1] > rospre[w + 1]
    const uint32_t base = t.rospre[w + 1u];
    if (t.rospre[g.nwin] == base) return false;
    uint32_t lo = w + 1u, hi = g.nwin - 1u;                 // answer in [lo, hi]
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (t.rospre[mid + 1u] > base) hi = mid;
        else lo = mid + 1u;
    }
    const uint32_t *bp = t.bp + t.nbase[lo];
    const uint32_t n = t.ccnt[lo];
    for (uint32_t j = 0; j < n; j++)
        if ((bp[j] >> 30) == kTrRos) {
            rw = lo;
            ri = j;
            return true;
        }
    return false;
}

AEC_HD uint64_t tr_G(const TrGeom &g, const TrTables &t, uint32_t w, uint32_t i)
{
    return t.gbase[w] + (t.bp[t.nbase[w] + i] & kTrBpMask);
}

// The jump: from node at `pos` with b blocks of the RSI done to the node in front of block `target` of the RSI
// (target = c.rsi: where the RSI ends; or a multiple of 64: where a segment starts -- a rest-of-segment run ends on
// one of those, so it never jumps over the target).
// kTrNone = unresolved (seam, end of the tables, zero run across the target, dead node on the way).
// end_at (optional): the node's record index (nbase + index inside its window).
AEC_HD uint64_t tr_jump_to(const Cfg &c, const TrGeom &g, const TrTables &t, uint64_t pos, uint32_t b, uint32_t target,
                           uint32_t *end_at = nullptr)
{
    uint32_t w, i;
    if (b > target || !tr_node_at(g, t, pos, w, i)) return kTrNone;
    if (b == target) {
        if (end_at) *end_at = t.nbase[w] + i;
        return pos;
    }
    const uint32_t seam0 = t.seampre[w];
    for (uint32_t guard = 0; guard <= c.rsi / 32u + 2u; guard++) {
        const uint32_t n = target - b;
        const uint64_t G = tr_G(g, t, w, i);
        uint32_t rw = 0, ri = 0;
        uint64_t dist = ~0ull;                       // blocks between this node and the next rest-of-segment run
        if (tr_next_ros(g, t, w, i, rw, ri)) dist = tr_G(g, t, rw, ri) - G;
        TR_DBG("jump: node w%u i%u G %llu b %u n %u nextros w%u i%u dist %lld\n", w, i, (unsigned long long)G, b, n, rw, ri, (long long)dist);
        if (dist >= n) {                             // the RSI ends in front of (or at) that run
            const uint32_t tw = tr_window_of(g, t, w, G + n);
            TR_DBG("  target window %u (nwin %u) seam %u/%u want %llu gbase %llu ccnt %u\n", tw, g.nwin,
                   tw < g.nwin ? t.seampre[tw] : 0u, seam0, (unsigned long long)(G + n),
                   (unsigned long long)(tw < g.nwin ? t.gbase[tw] : 0), tw < g.nwin ? t.ccnt[tw] : 0u);
            if (tw >= g.nwin && G + n == t.gbase[g.nwin]) {
                // the RSI ends where the trunk does: behind its last coded data set stands a node that covers nothing
                // (the end of the stream) -- the last RSI of a stream gets its record like any other
                uint32_t lw = g.nwin;
                while (lw > w && !t.ccnt[lw - 1u]) lw--;
                if (lw > w || t.ccnt[w]) {
                    const uint32_t ew = lw > w ? lw - 1u : w, ei = t.ccnt[ew] - 1u;
                    const uint32_t ebp = t.bp[t.nbase[ew] + ei];
                    if ((ebp >> 30) == kTrDead && t.gbase[ew] + (ebp & kTrBpMask) == G + n && t.seampre[ew] == seam0) {
                        if (end_at) *end_at = t.nbase[ew] + ei;
                        return g.lo + (uint64_t)ew * g.L + t.cpos[t.nbase[ew] + ei];
                    }
                }
                return kTrNone;
            }
            if (tw >= g.nwin || t.seampre[tw] != seam0) return kTrNone;
            uint32_t ti;
            if (!tr_node_of(g, t, tw, (uint32_t)(G + n - t.gbase[tw]), ti)) return kTrNone;
            if (end_at) *end_at = t.nbase[tw] + ti;
            return g.lo + (uint64_t)tw * g.L + t.cpos[t.nbase[tw] + ti];
        }
        if (t.seampre[rw] != seam0) return kTrNone;
        b += (uint32_t)dist;
        const uint32_t left_rsi = c.rsi - b, left_seg = 64u - (b % 64u);
        b += left_rsi < left_seg ? left_rsi : left_seg;
        // the node behind the run (it counts one block on the trunk)
        const uint64_t Gn = tr_G(g, t, rw, ri) + 1u;
        const uint32_t tw = tr_window_of(g, t, rw, Gn);
        if (tw >= g.nwin || t.seampre[tw] != seam0) return kTrNone;
        uint32_t ti;
        if (!tr_node_of(g, t, tw, (uint32_t)(Gn - t.gbase[tw]), ti)) return kTrNone;
        w = tw;
        i = ti;
        if (b == target) {
            if (end_at) *end_at = t.nbase[w] + i;
            return g.lo + (uint64_t)w * g.L + t.cpos[t.nbase[w] + i];
        }
        if (b > target) return kTrNone;
    }
    return kTrNone;
}

AEC_HD uint64_t tr_jump(const Cfg &c, const TrGeom &g, const TrTables &t, uint64_t pos, uint32_t b)
{
    return tr_jump_to(c, g, t, pos, b, c.rsi);
}

// ---- segment starts of a known RSI ------------------------------------------------------------------------
// The RSI that starts at `pos` (a true RSI start, found by the walk) and holds `nblocks` blocks (c.rsi, fewer for
// the RSI the input ends in): where do its segments of 64 blocks start?  The decoder can then take a lane per
// segment instead of one per RSI (aec_dec.hip: launch_decode_bare).  First part, tr_seg_walk: the coded data
// sets from the RSI start on, one by one -- the first with its reference sample -- until the walk stands on a
// node of the trunk; out(j, bit) for every segment start passed.  Second part: from that node (pos, b) every
// later segment start is ONE jump in the block numbering of the trunk, tr_jump_to(.., 64 j), independent of the
// others.  false = not to be had: a coded data set that does not parse, or a zero-block run across a segment
// border (legal for a decoder, reference decode.c:518-558; the reference's encoder ends runs there,
// encode.c:649) -- the decoder then takes the RSI as one item.
// Src: where bits and trunk marks come from (TrGlobal, or TrStaged: a stretch the caller has staged -- the walk
// then stops with TR_SEG_MORE when it leaves the stretch in front of `limit`, to be called again on the next one).
enum : uint32_t { TR_SEG_FAIL = 0, TR_SEG_ON_TRUNK = 1, TR_SEG_MORE = 2 };
template <class Src, class Out>
AEC_HD uint32_t tr_seg_walk(const TrStream &s, const Cfg &c, const Src &src, uint64_t &pos, uint32_t &b, uint32_t nblocks,
                            uint64_t limit, Out out)
{
    if (b == 0u) out(0u, pos);
    while (b < nblocks) {
        if (pos >= limit) return TR_SEG_MORE;
        if (b != 0u && src.marked(pos)) return TR_SEG_ON_TRUNK;
        const uint32_t ref = (b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
        uint32_t nz;
        TrWin W;
        src.win(pos, W);
        const uint32_t len = tr_cds(s, c, pos, ref, nz, W);
        if (!len) return TR_SEG_FAIL;
        const uint32_t nb = tr_blocks(c, nz, b);
        if (!nb || nb > nblocks - b || (b % 64u) + nb > 64u) return TR_SEG_FAIL;
        pos += len;
        b += nb;
        if ((b % 64u) == 0u && b < nblocks) out(b / 64u, pos);
    }
    return TR_SEG_ON_TRUNK;            // (all blocks walked: no segment is left to search for)
}

// jump (k_hyp_land): resolves a parked hypothesis
AEC_HD void tr_hyp_land(const Cfg &c, const TrGeom &g, const TrTables &t, uint32_t w, uint32_t idx)
{
    const uint64_t at = t.nbase[w] + idx;
    const uint32_t pk = t.park[at];
    if ((pk & kCoTag) && !(pk & kTrParked)) {               // a coalescing record (section 2b)
        if (co_kind(pk) != CO_LAND) {                       // (not finished: the plain walk of the node takes over)
            t.rec[at] = TrRec{0u, 0u};
            return;
        }
    } else if (!(pk & kTrParked)) {
        return;
    }
    const TrRec r = t.rec[at];
    const uint64_t c0 = g.lo + (uint64_t)w * g.L + t.cpos[at];
    const bool co = !(pk & kTrParked);
    const uint32_t k = co ? 0u : (pk >> 16) & 0x3Fu, b = co ? co_bend(pk) : pk & 0xFFFFu;
    uint32_t end_at = 0;
    const uint64_t e = tr_jump_to(c, g, t, c0 + r.x, b, c.rsi, &end_at);
    const uint32_t x = e == kTrNone ? 0u : tr_rec_pack(e - c0, k + 1u);
    // y: records of several RSIs -- the list of the RSI ends inside; a record of ONE RSI has no list, and says
    // instead which node it ends on (record index + 1), so that the walker's next lookup is a single read
    t.rec[at] = TrRec{x, !x ? 0u : (k == 0u ? end_at + 1u : r.y)};
}

// The RSI ends inside the record of a node (k RSIs, list head y): out(j, end) for the RSIs j = k - 2 .. 0,
// end = where RSI j of the record ends (absolute bit).  false = the list is broken (cannot happen).
template <class Out>
AEC_HD bool tr_rec_ends(const TrTables &t, uint64_t node_pos, uint32_t k, uint32_t y, Out out)
{
    for (uint32_t j = k - 1u; j-- > 0u;) {
        if (!y) return false;
        const TrPoolEntry e = t.pool[y - 1u];
        out(j, node_pos + e.end);
        y = e.prev;
    }
    return true;
}

// The RSI starts inside a record that covers k > 1 RSIs, parsed again from its node (the expansion of the
// records the true walk took): out[j] = start of RSI j + 1 of the record, j < k - 1.  false = the stream
// does not parse (cannot happen for a record that was built from it).  With AEC_PAD_RSI p and the values
// handed to `out` are where the RSIs in front END; tr_rsi_start() of them is where the next ones start.
template <class Out>
AEC_HD bool tr_rec_starts(const TrStream &s, const Cfg &c, uint64_t p, uint32_t k, Out out)
{
    uint32_t b = 0, done = 0;
    p = tr_rsi_start(c, p);
    while (done + 1u < k) {
        const uint32_t ref = (b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
        uint32_t nz;
        const uint32_t len = tr_cds(s, c, p, ref, nz);
        if (!len) return false;
        const uint32_t nb = tr_blocks(c, nz, b);
        if (!nb || nb > c.rsi - b) return false;
        p += len;
        b += nb;
        if (b == c.rsi) {
            out(done, p);
            done++;
            b = 0;
            p = tr_rsi_start(c, p);
        }
    }
    return true;
}

}  // namespace aec
