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
#define TR_DBG(...) do { if (g_dbg) fprintf(stderr, __VA_ARGS__); } while (0)
#include "../../libaec_amd/csrc/aec_trunk.h"
#include "../../libaec_amd/csrc/aec_cfg.h"

using namespace aec;

namespace {

struct Tables {
    std::vector<uint32_t> bitmap, bp, ccnt, nblk, nros, seampre, rospre, nbase;
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
        return TrTables{bitmap.data(), pre.data(), nbase.data(), cpos.data(), bp.data(), ccnt.data(), nblk.data(), nros.data(),
                        entry.data(), exit.data(), gbase.data(), seampre.data(), rospre.data(), rec.data()};
    }
};

}  // namespace

// stats[]: 0 windows, 1 seams, 2 nodes, 3 walk steps (sum), 4 walk steps (max), 5 landed, 6 done by walk,
// 7 failed walks, 8 unresolved jumps, 9 true starts checked, 10 true starts resolved, 11 true starts that are
// no node, 12 mismatches, 13 records covering > 1 RSI at true starts, 14 chain mismatches, 15 seams before repair,
// 16 rest-of-segment nodes, 17 hops of the table walk from start_bit, 18 RSIs it covered, 19 serial fallbacks
extern "C" int emul_trunk(const uint32_t *params, const uint8_t *enc, size_t enc_len, uint64_t start_bit,
                          uint32_t L, uint32_t lead, uint32_t budget, const uint64_t *offs, uint64_t n_offs,
                          uint64_t *stats, uint32_t rw, uint32_t passes, uint32_t capdiv)
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
    g.ncore = g.nwin;
    g.budget = budget;
    Tables T;
    TrTables t = T.view(g);
    for (int i = 0; i < 20; i++) stats[i] = 0;
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
    for (uint32_t w = 0; w < g.ncore; w++)
        for (uint32_t i = 0; i < t.ccnt[w]; i++) {
            TrHyp h;
            tr_hyp_start(c, h, g.lo + (uint64_t)w * L + t.cpos[t.nbase[w] + i]);
            uint32_t st;
            while ((st = tr_hyp_step(s, c, g, t, h)) == TR_RUN) {}
            stats[3] += h.steps;
            if (h.steps > stats[4]) stats[4] = h.steps;
            stats[5] += st == TR_LAND;
            stats[6] += st == TR_DONE;
            stats[7] += st == TR_FAIL;
            tr_hyp_finish(g, t, w, i, h, st);
        }
    for (uint32_t w = 0; w < g.ncore; w++)
        for (uint32_t i = 0; i < t.ccnt[w]; i++) {
            const bool parked = t.rec[t.nbase[w] + i].y & kTrParked;
            tr_hyp_land(c, g, t, w, i);
            if (parked && !t.rec[t.nbase[w] + i].x) stats[8]++;
        }
    for (uint32_t w = 0; w < g.ncore; w++)
        for (uint32_t i = 0; i < t.ccnt[w]; i++) tr_hyp_chain(g, t, w, i);

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
                while ((st = tr_hyp_step(s, c, g, t, h)) == TR_RUN) fprintf(stderr, "  step -> pos %llu b %u k %u\n", (unsigned long long)h.pos, h.b, h.k);
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
            if (!okk) stats[12]++;
        }
        if (rc.y) {
            const uint32_t cnt = rc.y >> 24, bits = rc.y & 0xFFFFFFu;
            if (r + cnt >= nk || key[r] + bits != key[r + cnt]) stats[14]++;
        } else {
            stats[14]++;
        }
    }
    // ---- the walk the serial walker would do: hop over records from start_bit, count fallbacks
    {
        uint64_t r = 0;
        while (r + 2 < nk) {
            uint32_t w, i;
            if (tr_node_at(g, t, key[r], w, i) && t.rec[t.nbase[w] + i].y && r + (t.rec[t.nbase[w] + i].y >> 24) < nk) {
                r += t.rec[t.nbase[w] + i].y >> 24;
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
