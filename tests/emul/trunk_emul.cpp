// trunk_emul.cpp -- CPU harness for libaec_amd/csrc/aec_trunk.h (TEST INFRASTRUCTURE).
// Runs the trunk index lane by lane (the kernels of aec_idx.hip are loops over these functions) over a
// whole stream and checks the records against the true RSI starts the caller passes in (from the oracle /
// the reference encoder).  Not linked into the product library.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

static int g_dbg = 0;
static unsigned long long g_cnt[4];
#define TR_COUNT_STEP(i) (g_cnt[i]++)
#define TR_DBG(...) do { if (g_dbg) fprintf(stderr, __VA_ARGS__); } while (0)
#include "../../libaec_amd/csrc/aec_trunk.h"
#include "../../libaec_amd/csrc/aec_cfg.h"

using namespace aec;

namespace {

struct Tables {
    std::vector<uint32_t> bitmap, bp, ccnt, nblk, nros, seampre, rospre, nbase, park, pool_cnt;
    std::vector<TrPoolEntry> pool;
    std::vector<uint16_t> pre, cpos;
    std::vector<uint64_t> entry, exit, exit2, gbase;
    std::vector<TrRec> rec;
    TrTables view(const TrGeom &g)
    {
        const size_t nw = g.L / 32;
        bitmap.assign((size_t)g.nwin * nw, 0);
        pre.assign((size_t)g.nwin * nw, 0);
        cpos.assign((size_t)g.ncap, 0);
        bp.assign((size_t)g.ncap, 0);
        nbase.assign(g.nwin + 1, 0);
        ccnt.assign(g.nwin, 0);
        nblk.assign(g.nwin, 0);
        nros.assign(g.nwin, 0);
        entry.assign(g.nwin, 0);
        exit.assign(g.nwin, 0);
        exit2.assign(g.nwin, 0);
        gbase.assign(g.nwin + 1, 0);
        seampre.assign(g.nwin + 1, 0);
        rospre.assign(g.nwin + 1, 0);
        rec.assign((size_t)g.ncap, TrRec{0, 0});
        park.assign((size_t)g.ncap, 0);
        pool.assign((size_t)g.pcap, TrPoolEntry{0, 0});
        pool_cnt.assign(1, 0);
        return TrTables{bitmap.data(), pre.data(), nbase.data(), cpos.data(), bp.data(), ccnt.data(), nblk.data(), nros.data(),
                        entry.data(), exit.data(), gbase.data(), seampre.data(), rospre.data(), rec.data(), park.data(),
                        pool.data(), pool_cnt.data()};
    }
};

}  // namespace

// stats[]: 0 windows, 1 seams, 2 nodes, 3 walk steps (sum), 4 walk steps (max), 5 landed, 6 done by walk,
// 7 failed walks, 8 unresolved jumps, 9 true starts checked, 10 true starts resolved, 11 true starts that are
// no node, 12 mismatches, 13 records covering > 1 RSI at true starts, 14 chain mismatches, 15 seams before repair,
// 16 rest-of-segment nodes, 17 hops of the table walk from start_bit, 18 RSIs it covered, 19 serial fallbacks
extern "C" int emul_trunk(const uint32_t *params, const uint8_t *enc, size_t enc_len, uint64_t start_bit,
                          uint32_t L, uint32_t lead, uint32_t budget, const uint64_t *offs, uint64_t n_offs,
                          uint64_t *stats, uint32_t rw, uint32_t passes, uint32_t capdiv, uint32_t staged)
{
    Cfg c;
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint32_t> words((enc_len + 3) / 4 + 1, 0);
    memcpy(words.data(), enc, enc_len);
    const TrStream s{words.data(), (enc_len + 3) / 4, (uint64_t)enc_len * 8};
    TrGeom g{};
    g.lo = start_bit / L * L;
    g.start_bit = start_bit;
    g.L = L;
    g.lead = lead;
    
    g.rw = rw ? rw : 1;
    g.nwin = (uint32_t)((s.end_bit - g.lo) / L + 1);
    g.ncap = (uint32_t)(((uint64_t)g.nwin * L) / (capdiv ? capdiv : 8));
    g.pcap = g.ncap;
    g.ncore = g.nwin;
    g.budget = budget;
    g.kmax = kTrMaxK;
    Tables T;
    TrTables t = T.view(g);
    for (int i = 0; i < 20; i++) stats[i] = 0;
    g_cnt[0] = g_cnt[1] = g_cnt[2] = 0;
    const uint32_t nreg = (g.nwin + g.rw - 1) / g.rw;
    uint64_t *ea = T.exit.data(), *eb = T.exit2.data();
    for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, nullptr, ea);
    for (uint32_t w = 1; w < g.nwin; w++) stats[15] += ea[w - 1] == kTrNone || ea[w - 1] != t.entry[w];
    for (uint32_t p = 0; p < passes; p++) {
        for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, ea, eb);
        std::swap(ea, eb);
    }
    t.exit = ea;
    tr_scan_serial(g, t);
    for (uint32_t w = 0; w < g.nwin; w++) tr_trunk_window(s, c, g, t, w, t.entry[w], nullptr, TR_FILL);
    if (getenv("TR_DEBUG")) {
        uint32_t empty = 0, dead = 0, noentry = 0;
        for (uint32_t w = 0; w < g.nwin; w++) {
            empty += t.ccnt[w] == 0;
            dead += t.exit[w] == kTrNone;
            noentry += t.entry[w] == kTrNone;
        }
        fprintf(stderr, "windows %u empty %u exit-none %u entry-none %u\n", g.nwin, empty, dead, noentry);
    }
    stats[0] = g.nwin;
    stats[1] = t.seampre[g.nwin] - 1;
    for (uint32_t w = 0; w < g.nwin; w++) {
        stats[2] += t.ccnt[w];
        stats[16] += t.nros[w];
    }
    auto alloc = [&]() -> uint32_t {
        if (t.pool_cnt[0] >= g.pcap) return 0xFFFFFFFFu;
        return t.pool_cnt[0]++;
    };
    const TrGlobal mem{s, g, t};
    // staged == 0: every walk reads device memory.  Else: the walks of each group of `staged` windows run on a
    // staged stretch (group + one more window), as the workgroups of k_hyp_walk_lds do.
    std::vector<uint32_t> sw, bm;
    std::vector<uint8_t> nx;
    const bool no_table = staged >= 100;                  // (staged stretches without the byte table)
    if (no_table) staged -= 100;
    for (uint32_t w0 = 0; w0 < g.ncore; w0 += (staged ? staged : g.ncore)) {
        const uint32_t w1 = staged ? (w0 + staged < g.ncore ? w0 + staged : g.ncore) : g.ncore;
        TrStaged st{mem, nullptr, nullptr, nullptr, 0, 0};
        if (staged) {
            st.base = g.lo + (uint64_t)w0 * L;
            st.bits = (w1 - w0 + 1) * L;
            sw.assign(st.bits / 32 + 8, 0);
            bm.assign(st.bits / 32, 0);
            nx.assign(st.bits, 0);
            for (uint32_t i = 0; i < st.bits / 32 + 8; i++) sw[i] = tr_word(s, (st.base >> 5) + i);
            for (uint32_t i = 0; i < st.bits / 32; i++) {
                const uint64_t gw = (st.base - g.lo) / 32 + i;
                bm[i] = gw < (uint64_t)g.nwin * (L / 32) ? t.bitmap[gw] : 0u;
            }
            st.sw = sw.data();
            st.bm = bm.data();
            st.nx = no_table ? nullptr : nx.data();
            for (uint32_t q = 0; q < st.bits && !no_table; q++) {
                TrWin W;
                st.win(st.base + q, W);
                nx[q] = tr_fast_entry(s, c, st.base + q, W);
            }
        }
        for (uint32_t w = w0; w < w1; w++)
            for (uint32_t i = 0; i < t.ccnt[w]; i++) {
                TrHyp h;
                tr_hyp_start(c, h, g.lo + (uint64_t)w * L + t.cpos[t.nbase[w] + i]);
                uint32_t stt;
                auto commit = [&]() {                    // (false: the pool is full, the hypothesis fails)
                    return !h.pend || tr_hyp_commit(g, t, h, alloc());
                };
                if (staged) {
                    while ((stt = tr_hyp_step(s, c, g, st, h)) == TR_RUN)
                        if (!commit()) {
                            stt = TR_FAIL;
                            break;
                        }
                } else {
                    while ((stt = tr_hyp_step(s, c, g, mem, h)) == TR_RUN)
                        if (!commit()) {
                            stt = TR_FAIL;
                            break;
                        }
                }
                stats[3] += h.steps;
                if (h.steps > stats[4]) stats[4] = h.steps;
                if (const char *hp = getenv("TR_HIST")) {      // (diagnostics: walk lengths, one per line)
                    static FILE *hf = fopen(hp, "w");
                    if (hf) fprintf(hf, "%u %u %u\n", w, h.steps, (unsigned)stt);
                }
                stats[5] += stt == TR_LAND;
                stats[6] += stt == TR_DONE;
                stats[7] += stt == TR_FAIL;
                tr_hyp_finish(g, t, w, i, h, stt);
            }
    }
    if (getenv("TR_DEBUG")) fprintf(stderr, "steps: fast %llu, parse inside the stretch %llu, parse outside %llu\n", g_cnt[0], g_cnt[1], g_cnt[2]);
    for (uint32_t w = 0; w < g.ncore; w++)
        for (uint32_t i = 0; i < t.ccnt[w]; i++) {
            const bool parked = t.park[t.nbase[w] + i] & kTrParked;
            tr_hyp_land(c, g, t, w, i);
            if (parked && !t.rec[t.nbase[w] + i].x) stats[8]++;
        }

    // ---- truth: the serial walk with this header's own parser, checked against the caller's RSI starts
    // (offs[0 .. n_offs - 1) = starts from the oracle, offs[n_offs - 1] = end of the stream's last CDS).
    // key[r] = where the records of RSI r are looked up: its start, with AEC_PAD_RSI the end of RSI r - 1.
    int bad = 0;
    std::vector<uint64_t> key;
    {
        uint64_t p = start_bit;
        key.push_back(p);
        for (uint64_t r = 0; r + 1 < n_offs; r++) {
            uint64_t q = tr_rsi_start(c, p);
            if (q != offs[r]) {
                if (bad++ < 5) fprintf(stderr, "serial walk: RSI %llu starts at %llu, oracle %llu\n",
                                       (unsigned long long)r, (unsigned long long)q, (unsigned long long)offs[r]);
                return 2;
            }
            if (r + 2 == n_offs) break;                    // (the last RSI may be short: no end to find)
            uint32_t b = 0;
            while (b < c.rsi) {
                uint32_t nz;
                const uint32_t len = tr_cds(s, c, q, (b == 0 && (c.flags & F_PREPROCESS)) ? 1u : 0u, nz);
                const uint32_t nb = len ? tr_blocks(c, nz, b) : 0;
                if (!len || !nb || nb > c.rsi - b) return 3;
                q += len;
                b += nb;
            }
            p = q;
            key.push_back(p);
        }
    }
    // (a record may also end where the stream does: the last RSI, complete or closed by a rest-of-segment run
    // that no decoder can tell from one reaching the nominal end)
    key.push_back(offs[n_offs - 1]);
    const uint64_t nk = key.size();                        // key[nk - 2] starts the last RSI, key[nk - 1] ends it
    for (uint64_t r = 0; r + 2 < nk; r++) {
        stats[9]++;
        uint32_t w, i;
        if (!tr_node_at(g, t, key[r], w, i)) {
            stats[11]++;
            continue;
        }
        const TrRec rc = t.rec[t.nbase[w] + i];
        if (!rc.x && getenv("TR_VERBOSE")) {
            TrHyp h;
            tr_hyp_start(c, h, key[r]);
            uint32_t st;
            while ((st = tr_hyp_step(s, c, g, mem, h)) == TR_RUN && (!h.pend || tr_hyp_commit(g, t, h, alloc()))) {}
            fprintf(stderr, "unresolved true start rsi %llu at %llu: state %u pos %llu b %u k %u steps %u\n",
                    (unsigned long long)r, (unsigned long long)key[r], st, (unsigned long long)h.pos, h.b, h.k, h.steps);
            g_dbg = 1;
            if (st == TR_LAND) fprintf(stderr, "  jump -> %lld\n", (long long)tr_jump(c, g, t, h.pos, h.b));
            g_dbg = 0;
        }
        if (!rc.x) continue;
        stats[10]++;
        const uint32_t k = tr_rec_k(rc.x);
        if (k > 1) stats[13]++;
        if (r + k >= nk || key[r] + tr_rec_bits(rc.x) != key[r + k]) {
            if (bad++ < 5)
                fprintf(stderr, "mismatch at rsi %llu: key %llu rec bits %u k %u, true next %llu\n",
                        (unsigned long long)r, (unsigned long long)key[r], tr_rec_bits(rc.x), k,
                        (unsigned long long)(r + k < nk ? key[r + k] : 0));
            stats[12]++;
            if (getenv("TR_VERBOSE")) {
                TrHyp h;
                tr_hyp_start(c, h, key[r]);
                uint32_t st;
                while ((st = tr_hyp_step(s, c, g, mem, h)) == TR_RUN && (!h.pend || tr_hyp_commit(g, t, h, alloc()))) fprintf(stderr, "  step -> pos %llu b %u k %u\n", (unsigned long long)h.pos, h.b, h.k);
                fprintf(stderr, "  state %u pos %llu b %u k %u\n", st, (unsigned long long)h.pos, h.b, h.k);
                g_dbg = 1;
                if (st == TR_LAND) fprintf(stderr, "  jump -> %lld\n", (long long)tr_jump(c, g, t, h.pos, h.b));
                g_dbg = 0;
                for (uint32_t ww = 0; ww < g.nwin; ww++) {
                    fprintf(stderr, "  w%u gbase %llu seampre %u rospre %u entry %lld exit %lld:", ww, (unsigned long long)t.gbase[ww], t.seampre[ww], t.rospre[ww], (long long)t.entry[ww], (long long)t.exit[ww]);
                    for (uint32_t ii = 0; ii < t.ccnt[ww]; ii++) fprintf(stderr, " %llu/%u/%u", (unsigned long long)(g.lo + (uint64_t)ww * L + t.cpos[t.nbase[ww] + ii]), t.bp[t.nbase[ww] + ii] & kTrBpMask, t.bp[t.nbase[ww] + ii] >> 30);
                    fprintf(stderr, "\n");
                }
            }
            continue;
        }
        if (k > 1) {
            const uint64_t r0 = r;
            const bool okk = tr_rec_starts(s, c, key[r], k, [&](uint32_t j, uint64_t q) {
                if (q != key[r0 + j + 1]) stats[12]++;
            });
            const bool okl = tr_rec_ends(t, key[r], k, rc.y, [&](uint32_t j, uint64_t q) {
                if (q != key[r0 + j + 1]) stats[12]++;
            });
            if (!okk || !okl) stats[12]++;
        }
    }
    // the LAST RSI: when it is a whole one its record ends where the stream does (on the node behind the last coded
    // data set, which covers nothing); a record there must say exactly that
    if (nk >= 2) {
        uint32_t w, i;
        if (tr_node_at(g, t, key[nk - 2], w, i)) {
            const TrRec rc = t.rec[t.nbase[w] + i];
            if (rc.x && (tr_rec_k(rc.x) != 1u || key[nk - 2] + tr_rec_bits(rc.x) != key[nk - 1])) {
                fprintf(stderr, "last RSI: record of %u RSIs over %u bits, the stream ends %llu bits behind its start\n",
                        tr_rec_k(rc.x), tr_rec_bits(rc.x), (unsigned long long)(key[nk - 1] - key[nk - 2]));
                stats[12]++;
            }
            if (getenv("TR_DEBUG")) fprintf(stderr, "last RSI: %s\n", rc.x ? "resolved" : "no record");
        }
    }
    // ---- the walk the serial walker would do: hop over records from start_bit, count fallbacks
    {
        uint64_t r = 0;
        while (r + 2 < nk) {
            uint32_t w, i;
            if (tr_node_at(g, t, key[r], w, i) && t.rec[t.nbase[w] + i].x && r + tr_rec_k(t.rec[t.nbase[w] + i].x) < nk) {
                r += tr_rec_k(t.rec[t.nbase[w] + i].x);
                stats[17]++;
            } else {
                stats[19]++;
                r++;
            }
        }
        stats[18] = r;
    }
    if (stats[12] || stats[14]) bad++;
    return bad ? 1 : 0;
}

// Segment starts of every RSI (aec_trunk.h: tr_seg_walk + tr_jump_to, the lane functions of k_seg_starts) against
// the serial walk.  stats[]: 0 RSIs checked, 1 RSIs whose segment starts all came out, 2 segment starts that differ
// from the serial walk (must be 0), 3 RSIs given up (zero run across a segment border, end of the tables),
// 4 segments checked, 5 coded data sets the walks parsed before they stood on the trunk (sum)
extern "C" int emul_segments(const uint32_t *params, const uint8_t *enc, size_t enc_len, uint32_t L, uint32_t lead,
                             uint32_t rw, uint32_t passes, const uint64_t *offs, uint64_t n_offs, uint64_t *stats)
{
    Cfg c;
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint32_t> words((enc_len + 3) / 4 + 1, 0);
    memcpy(words.data(), enc, enc_len);
    const TrStream s{words.data(), (enc_len + 3) / 4, (uint64_t)enc_len * 8};
    TrGeom g{};
    g.lo = 0;
    g.start_bit = 0;
    g.L = L;
    g.lead = lead;
    g.rw = rw ? rw : 1;
    g.nwin = (uint32_t)(s.end_bit / L + 1);
    g.ncap = (uint32_t)(((uint64_t)g.nwin * L) / 8);
    g.pcap = g.ncap;
    g.ncore = g.nwin;
    g.budget = 65536;
    g.kmax = kTrMaxK;
    Tables T;
    TrTables t = T.view(g);
    for (int i = 0; i < 6; i++) stats[i] = 0;
    const uint32_t nreg = (g.nwin + g.rw - 1) / g.rw;
    uint64_t *ea = T.exit.data(), *eb = T.exit2.data();
    for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, nullptr, ea);
    for (uint32_t p = 0; p < passes; p++) {
        for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, ea, eb);
        std::swap(ea, eb);
    }
    t.exit = ea;
    tr_scan_serial(g, t);
    for (uint32_t w = 0; w < g.nwin; w++) tr_trunk_window(s, c, g, t, w, t.entry[w], nullptr, TR_FILL);
    int bad = 0;
    for (uint64_t r = 0; r + 1 < n_offs; r++) {           // offs[n_offs - 1] = end of the stream
        // truth: the serial walk of this RSI
        std::vector<uint64_t> truth;
        uint64_t q = offs[r];
        uint32_t b = 0;
        bool clean = true;                                 // no zero run across a segment border
        truth.push_back(q);
        while (b < c.rsi && q < offs[r + 1]) {
            uint32_t nz;
            const uint32_t len = tr_cds(s, c, q, (b == 0 && (c.flags & F_PREPROCESS)) ? 1u : 0u, nz);
            const uint32_t nb = len ? tr_blocks(c, nz, b) : 0;
            if (!len || !nb || nb > c.rsi - b) return 3;
            if ((b % 64u) + nb > 64u) clean = false;
            q += len;
            b += nb;
            if ((b % 64u) == 0u && b < c.rsi && q < offs[r + 1]) truth.push_back(q);
        }
        const uint32_t nblocks = b;
        stats[0]++;
        std::vector<uint64_t> got((nblocks + 63u) / 64u, ~0ull);
        uint64_t pos = offs[r];
        uint32_t bw = 0;
        unsigned steps = 0;
        const TrGlobal mem{s, g, t};
        bool ok = tr_seg_walk(s, c, mem, pos, bw, nblocks, ~0ull, [&](uint32_t j, uint64_t p) { got[j] = p; }) == TR_SEG_ON_TRUNK;
        (void)steps;
        if (ok)
            for (uint32_t j = bw / 64u + 1u; j < got.size(); j++) {
                const uint64_t e = tr_jump_to(c, g, t, pos, bw, j * 64u);
                if (e == kTrNone) ok = false;
                got[j] = e;
            }
        stats[5] += bw;
        if (!ok) {
            stats[3]++;
            continue;
        }
        if (!clean) bad++;                                 // (must have been given up)
        stats[1]++;
        for (size_t j = 0; j < got.size(); j++) {
            stats[4]++;
            if (j >= truth.size() || got[j] != truth[j]) {
                if (stats[2]++ < 5)
                    fprintf(stderr, "segment %zu of RSI %llu: %llu, serial walk %llu\n", j, (unsigned long long)r,
                            (unsigned long long)got[j], (unsigned long long)(j < truth.size() ? truth[j] : 0));
            }
        }
    }
    return (bad || stats[2]) ? 1 : 0;
}

// Coalescing hypothesis walks (aec_trunk.h section 2b: the lane functions of k_hyp_walk_co / k_hyp_walk_rest /
// k_hyp_defer), group by group as the workgroups take them (one walk after the other here), against the plain walk
// of every node (tr_hyp_step): where the coalescing walk gives a landing (node + block count) it must be the plain
// walk's.  stats[]: 0 nodes, 1 landed inside the group, 2 handed on (roots), 3 waited for a root and resolved,
// 4 left to the plain walk, 5 mismatches (must be 0), 6 parses of the coalescing walks, 7 parses of the handed-on
// walks, 8 parses the plain walks take for the same nodes, 9 nodes whose plain walk does not land (k > 0, failed)
extern "C" int emul_coalesce(const uint32_t *params, const uint8_t *enc, size_t enc_len, uint32_t L, uint32_t lead,
                             uint32_t rw, uint32_t passes, uint32_t wpg, uint32_t margin, uint32_t shift, uint32_t tmax,
                             uint64_t *stats, const uint64_t *offs, uint64_t n_offs)
{
    // (offs, optional: the true RSI starts -- every one of them that is a node and whose plain walk lands must get its
    // landing from the coalescing walks as well; those that do not are reported)
    Cfg c;
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint32_t> words((enc_len + 3) / 4 + 1, 0);
    memcpy(words.data(), enc, enc_len);
    const TrStream s{words.data(), (enc_len + 3) / 4, (uint64_t)enc_len * 8};
    TrGeom g{};
    g.lo = 0;
    g.start_bit = 0;
    g.L = L;
    g.lead = lead;
    g.rw = rw ? rw : 1;
    g.nwin = (uint32_t)(s.end_bit / L + 1);
    g.ncap = (uint32_t)(((uint64_t)g.nwin * L) / 8);
    g.pcap = g.ncap;
    g.ncore = g.nwin;
    g.budget = 65536;
    g.kmax = kTrMaxK;
    Tables T;
    TrTables t = T.view(g);
    for (int i = 0; i < 10; i++) stats[i] = 0;
    const uint32_t nreg = (g.nwin + g.rw - 1) / g.rw;
    uint64_t *ea = T.exit.data(), *eb = T.exit2.data();
    for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, nullptr, ea);
    for (uint32_t p = 0; p < passes; p++) {
        for (uint32_t r = 0; r < nreg; r++) tr_trunk_region(s, c, g, t, r, ea, eb);
        std::swap(ea, eb);
    }
    t.exit = ea;
    tr_scan_serial(g, t);
    for (uint32_t w = 0; w < g.nwin; w++) tr_trunk_window(s, c, g, t, w, t.entry[w], nullptr, TR_FILL);
    const TrGlobal mem{s, g, t};
    struct Cells {
        std::vector<uint32_t> v;
        uint32_t claim(uint32_t i, uint32_t p)
        {
            const uint32_t old = v[i];
            if (!old) v[i] = p;
            return old;
        }
        uint32_t peek(uint32_t i) const { return v[i]; }
    } cells;
    int bad = 0;
    for (uint32_t w0 = 0; w0 < g.ncore; w0 += wpg) {
        const uint32_t w1 = w0 + wpg < g.ncore ? w0 + wpg : g.ncore;
        const uint64_t base = (uint64_t)w0 * L;
        const uint32_t bits = (w1 - w0) * L + margin;
        std::vector<uint32_t> sw(bits / 32 + 8), bm(bits / 32);
        for (uint32_t i = 0; i < bits / 32 + 8; i++) sw[i] = tr_word(s, (base >> 5) + i);
        for (uint32_t i = 0; i < bits / 32; i++) {
            const uint64_t gw = base / 32 + i;
            bm[i] = gw < (uint64_t)g.nwin * (L / 32) ? t.bitmap[gw] : 0u;
        }
        const TrStaged st{mem, sw.data(), bm.data(), nullptr, base, bits};
        cells.v.assign((bits >> shift) + 1, 0);
        std::vector<uint64_t> npos;
        for (uint32_t w = w0; w < w1; w++)
            for (uint32_t i = 0; i < t.ccnt[w]; i++) npos.push_back((uint64_t)w * L + t.cpos[t.nbase[w] + i]);
        const uint32_t n = (uint32_t)npos.size();
        std::vector<CoRec> recs(n);
        // the walks (an arbitrary order stands for the concurrency of the lanes: pseudo-random)
        std::vector<uint32_t> ord(n);
        for (uint32_t i = 0; i < n; i++) ord[i] = i;
        uint64_t x = 88172645463325252ull + w0;
        for (uint32_t i = n; i > 1; i--) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(ord[i - 1], ord[x % i]);
        }
        const uint64_t lim = base + bits - 1024;
        for (uint32_t oi = 0; oi < n; oi++) {
            const uint32_t m = ord[oi];
            if (m >= kCoMaxOwners) {
                recs[m] = CoRec{co_pack(CO_PLAIN, 0, kCoNoRos), 0};
                continue;
            }
            CoWalk h;
            tr_co_start(c, h, npos[m]);
            uint32_t r, hit = 0;
            while ((r = tr_co_step(s, c, st, cells, h, m, base, lim, shift, tmax, hit)) == CO_RUN) {}
            stats[6] += h.steps;
            uint32_t tt = 0;
            if (r == CO_LAND || r == CO_QUEUE || r == CO_OVER) {
                if (h.pos - h.c > 0xFFFFFFFFull) r = CO_PLAIN;
                tt = (uint32_t)(h.pos - h.c);
            } else if (r == CO_LINK) {
                tt = hit;
            }
            recs[m] = CoRec{co_pack(r, h.b, h.b_ros), tt};
        }
        // resolution inside the group; roots are finished from memory; guests of roots afterwards
        std::vector<uint32_t> fin_kind(n), fin_b(n), fin_root(n);
        std::vector<uint64_t> fin_z(n);
        std::vector<uint32_t> root_k(n, 0), root_dist(n, 0);
        for (uint32_t m = 0; m < n; m++) {
            uint64_t z = 0;
            uint32_t root = 0, b = 0;
            fin_kind[m] = tr_co_resolve(c, [&](uint32_t q) { return recs[q]; }, [&](uint32_t q) { return npos[q]; }, m, z, root, b);
            fin_z[m] = z;
            fin_root[m] = root;
            fin_b[m] = b;
        }
        for (uint32_t m = 0; m < n; m++)
            if (fin_kind[m] == CO_GOON) {                  // (on from where another walk ran over, as a root of its own)
                uint32_t dist = 0, parses = 0;
                root_k[m] = tr_co_rest(s, c, g, mem, npos[m], fin_z[m], fin_b[m], kCoNoRos, dist, &parses);
                root_dist[m] = dist;
                stats[7] += parses;
                fin_kind[m] = CO_QUEUE;
            } else if (fin_kind[m] == CO_QUEUE) {
                uint32_t dist = 0;
                // (count the parses of the rest walk: by the plain parser's counter -- not available; by re-walking)
                uint32_t parses = 0;
                root_k[m] = tr_co_rest(s, c, g, mem, npos[m], npos[m] + recs[m].t, co_bend(recs[m].k), co_bros(recs[m].k), dist, &parses);
                stats[7] += parses;
                root_dist[m] = dist;
                stats[2]++;
            }
        for (uint32_t m = 0; m < n; m++) {
            stats[0]++;
            // the plain walk of this node
            TrHyp h;
            tr_hyp_start(c, h, npos[m]);
            uint32_t stt;
            while ((stt = tr_hyp_step(s, c, g, mem, h)) == TR_RUN) h.pend = 0;
            stats[8] += h.steps;
            const bool plain_lands = stt == TR_LAND && h.k == 0;
            if (!plain_lands) stats[9]++;
            bool have = false;
            uint64_t z = 0;
            uint32_t b = 0;
            if (fin_kind[m] == CO_LAND) {
                have = true;
                z = fin_z[m];
                b = fin_b[m];
                stats[1]++;
            } else if (fin_kind[m] == CO_QUEUE) {
                if (co_kind(root_k[m]) == CO_LAND) {
                    have = true;
                    z = npos[m] + root_dist[m];
                    b = co_bend(root_k[m]);
                }
            } else if (fin_kind[m] == CO_DEFER) {
                const uint32_t R = fin_root[m];
                uint32_t dk;
                const uint32_t bb = co_defer(c, fin_b[m], co_bend(recs[R].k), root_k[R], dk);
                if (bb) {
                    have = true;
                    z = npos[R] + root_dist[R];
                    b = bb;
                    stats[3]++;
                }
            }
            if (!have && plain_lands && offs) {
                for (uint64_t q = 0; q < n_offs; q++)
                    if (offs[q] == npos[m])
                        fprintf(stderr, "true RSI start %llu (node %llu, group %u local %u): no landing from the coalescing walks: kind %u own %u b %u root %u\n",
                                (unsigned long long)q, (unsigned long long)npos[m], w0, m, fin_kind[m], co_kind(recs[m].k), fin_b[m], fin_root[m]);
            }
            if (!have) {
                stats[4]++;
                // no landing: a CO_FAIL must be a plain walk that fails as well, a CO_OVER one that does not land in
                // its first RSI
                uint32_t why = fin_kind[m];
                if (why == CO_QUEUE) why = co_kind(root_k[m]);
                if (why == CO_DEFER) {
                    uint32_t dk;
                    (void)co_defer(c, fin_b[m], co_bend(recs[fin_root[m]].k), root_k[fin_root[m]], dk);
                    why = dk;
                }
                if ((why == CO_FAIL && stt != TR_FAIL) || (why == CO_OVER && plain_lands)) {
                    if (stats[5]++ < 8)
                        fprintf(stderr, "node %llu: coalescing says %u, plain walk: state %u b %u k %u\n", (unsigned long long)npos[m], why, stt, h.b, h.k);
                    bad = 1;
                }
                continue;
            }
            if (!plain_lands || z != h.pos || b != h.b) {
                if (stats[5]++ < 8)
                    fprintf(stderr, "node %llu (group %u, local %u, kind %u): coalesced lands at %llu with %u blocks, plain walk: state %u pos %llu b %u k %u\n",
                            (unsigned long long)npos[m], w0, m, fin_kind[m], (unsigned long long)z, b, stt, (unsigned long long)h.pos, h.b, h.k);
                bad = 1;
            }
        }
    }
    return bad;
}
