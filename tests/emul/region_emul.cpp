// region_emul.cpp -- the region index (libaec_amd/csrc/aec_region.h, aec_region.hip) on the CPU:
// the per-lane functions of the scheme, run region by region in a loop, against the RSI starts the oracle's encoder
// reports.  Two things are pinned here: (1) how often the GUESS of a region's entry is right on the benchmark shapes
// and on the reference's sample file (a wrong guess is caught and walked again in the product -- this is about speed),
// and (2) that the whole pass -- guesses, walks, the check of every entry against the walk in front, repairs, the RSI
// starts written -- delivers exactly the oracle's table whatever the guesses were.
// (test infrastructure; built by tests/test_region_emul.py)
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../libaec_amd/csrc/aec_cfg.h"
#include "../../libaec_amd/csrc/aec_region.h"

using namespace aec;

namespace {
struct HostRing {                       // the lane's ring as plain memory (stride 1)
    std::vector<uint32_t> mem;
    RgRingT<1u, 64u> ps;
    HostRing(const TrStream &s, const Cfg &c) : mem(rg_ring_rows(64u), 0u), ps{s, c} { ps.init(mem.data(), 5u); }
};
}  // namespace


extern "C" {

// params: bps, bs, rsi, flags.  stats: regions, right, none, wrong, parses, anchors, suspicions, lost, sum of
// (entry - region start) over the right ones
int emul_region_guess(const uint32_t *params, const uint8_t *stream, size_t nbytes, const uint64_t *rsi_off, size_t n_off,
                      uint64_t region_bits, uint32_t budget, uint64_t *stats, uint64_t *guesses)
{
    Cfg c{};
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint32_t> words((nbytes + 3) / 4 + 16, 0u);
    std::memcpy(words.data(), stream, nbytes);
    const TrStream s{words.data(), (nbytes + 3) / 4, (uint64_t)nbytes * 8};
    HostRing hr(s, c);
    std::vector<uint64_t> offs(rsi_off, rsi_off + n_off);
    std::sort(offs.begin(), offs.end());
    uint64_t regions = 0, right = 0, none = 0, wrong = 0, dist = 0, parses = 0;
    for (uint64_t from = region_bits; from + 64 < s.end_bit; from += region_bits) {
        regions++;
        uint64_t at = 0;
        const bool got = rg_guess(hr.ps, c, from, (uint32_t)(2 * region_bits), budget, at, &parses);
        if (guesses) guesses[regions - 1] = got ? at : ~0ull;
        if (!got) {
            none++;
            continue;
        }
        if (std::binary_search(offs.begin(), offs.end(), at)) {
            right++;
            dist += at - from;
        } else {
            wrong++;
        }
    }
    stats[0] = regions;
    stats[1] = right;
    stats[2] = none;
    stats[3] = wrong;
    stats[4] = parses;
    stats[5] = 0;
    stats[6] = 0;
    stats[7] = 0;
    stats[8] = dist;
    return 0;
}

// RgRing::cds (the lane's ring) against tr_cds (from memory) at EVERY bit of the stream, without and with a reference
// sample, in the order a walk would ask (increasing positions) and jumping about as the tests of the guess do; returns the
// first bit where they differ + 1, 0 = none
uint64_t emul_region_parser(const uint32_t *params, const uint8_t *stream, size_t nbytes, uint64_t stride)
{
    Cfg c{};
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return ~0ull;
    std::vector<uint32_t> words((nbytes + 3) / 4 + 16, 0u);
    std::memcpy(words.data(), stream, nbytes);
    const TrStream s{words.data(), (nbytes + 3) / 4, (uint64_t)nbytes * 8};
    RgMemParser pm{s, c};
    HostRing hr(s, c);
    hr.ps.seat(0);
    std::vector<uint32_t> mem32(rg_ring_rows(32u), 0u);      // (the walks' smaller ring beside it)
    RgRingT<1u, 32u> small{s, c};
    small.init(mem32.data(), 3u);
    small.seat(0);
    auto same = [&](uint64_t q, uint32_t ref) {
        uint32_t id0, nz0, id2, nz2, id3, nz3;
        const uint32_t l0 = pm.cds(q, ref, id0, nz0), l2 = hr.ps.cds(hr.ps.rel_of(q), ref, id2, nz2);
        if (small.base_bits != hr.ps.base_bits) small.seat(hr.ps.base_bits);
        const uint32_t l3 = small.cds(small.rel_of(q), ref, id3, nz3);
        return l0 == l2 && (!l0 || (id0 == id2 && nz0 == nz2)) && l0 == l3 && (!l0 || (id0 == id3 && nz0 == nz3));
    };
    for (uint64_t q = 0; q < s.end_bit; q += stride) {
        for (uint32_t ref = 0; ref < 2; ref++)
            if (!same(q, ref)) return q + 1;
        if ((q & 1023u) == 0u) {
            for (uint64_t d : {700ull, 300ull, 5000ull}) {
                const uint64_t back = q > d ? q - d : 0;
                if (back < hr.ps.base_bits) continue;    // (a ring serves positions from its base on)
                if (!same(back, 1u)) return back + 1;
            }
        }
        if ((q & 16383u) == 8191u) hr.ps.seat(q + 1);    // (a new base, as every walk has its own)
    }
    return 0;
}

// The whole pass as the kernels run it (aec_region.hip), region by region in loops: guesses, which of them are kept,
// walks, the check of every entry against the walk in front, repair passes, the count, the RSI starts.
// sabotage_every != 0: every such guess is moved by sabotage_shift bits (a wrong guess that the pass must mend).
// stats: regions, kept, entries that differ from the walk in front after the first walk, repair passes that had work,
// delivered, n_rsi, tail_blocks, end_bit
int emul_region_index(const uint32_t *params, const uint8_t *stream, size_t nbytes, uint64_t region_bits, uint32_t budget,
                      uint32_t passes, uint32_t sabotage_every, int64_t sabotage_shift, uint64_t *rsi_off, uint64_t cap,
                      uint64_t *stats)
{
    Cfg c{};
    if (make_cfg(params[0], params[1], params[2], params[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint32_t> words((nbytes + 3) / 4 + 16, 0u);
    std::memcpy(words.data(), stream, nbytes);
    const TrStream s{words.data(), (nbytes + 3) / 4, (uint64_t)nbytes * 8};
    const uint64_t nreg = (s.end_bit + region_bits - 1) / region_bits;
    std::vector<RgEntry> found(nreg), entry(nreg);
    found[0] = RgEntry{0, 0, 1};
    for (uint64_t r = 1; r < nreg; r++) {
        HostRing hr(s, c);
        uint64_t at = 0;
        const uint64_t from = r * region_bits;
        const bool got = rg_guess(hr.ps, c, from, (uint32_t)(2 * region_bits), budget, at);
        found[r] = RgEntry{got ? at : 0, 0, got ? 1u : 0u};
        if (got && sabotage_every && r % sabotage_every == 0) found[r].pos = (uint64_t)((int64_t)at + sabotage_shift);
    }
    uint64_t kept = 0;
    for (uint64_t r = 0; r < nreg; r++) {
        entry[r] = found[r];
        entry[r].live = (r == 0 || rg_keep(found[r], r + 1 < nreg ? found[r + 1] : RgEntry{}, r + 1 < nreg)) ? 1u : 0u;
        kept += entry[r].live;
    }
    auto next_live = [&](uint64_t r) {
        for (uint64_t q = r + 1; q < nreg; q++)
            if (entry[q].live) return q;
        return nreg;
    };
    auto prev_live = [&](uint64_t r) {
        uint64_t q = r;
        while (q-- > 0)
            if (entry[q].live) return q;
        return (uint64_t)0;
    };
    const uint64_t max_bits = 16 * region_bits;
    std::vector<RgState> ex(nreg);
    std::vector<uint32_t> cnt(nreg, 0);
    auto walk = [&](uint64_t r) {
        HostRing hr(s, c);
        auto &ps = hr.ps;
        RgState x{entry[r].pos, entry[r].b, 0};
        const uint64_t nl = next_live(r);
        uint32_t n = 0;
        rg_walk(ps, c, x, nl < nreg ? entry[nl].pos : ~0ull, nl < nreg ? max_bits : ~0ull,
                [&](uint64_t) { n++; return true; }, [](uint32_t, uint64_t) {});
        cnt[r] = n;
        ex[r] = x;
    };
    for (uint64_t r = 0; r < nreg; r++)
        if (entry[r].live) walk(r);
    auto differs = [&](uint64_t r) {          // (r live, r >= 1)
        const RgState &p = ex[prev_live(r)];
        return p.st != 0u || p.pos != entry[r].pos || p.b != entry[r].b;
    };
    uint64_t mism0 = 0, busy_passes = 0;
    for (uint64_t r = 1; r < nreg; r++) mism0 += entry[r].live && differs(r);
    for (uint32_t k = 0; k < passes; k++) {
        // (a snapshot, as the kernel's double buffers: which regions are mended is decided on the state before the pass)
        std::vector<uint64_t> todo;
        for (uint64_t r = 1; r < nreg; r++) {
            if (!entry[r].live || !differs(r)) continue;
            const uint64_t pr = prev_live(r);
            if (ex[pr].st) continue;                          // (the walk in front ended: nothing to enter on)
            if (pr != 0 && differs(pr)) continue;             // (the region in front is in doubt itself)
            todo.push_back(r);
        }
        if (todo.empty()) break;
        busy_passes++;
        std::vector<RgEntry> ne(todo.size());
        for (size_t i = 0; i < todo.size(); i++) {
            const RgState &p = ex[prev_live(todo[i])];
            ne[i] = RgEntry{p.pos, p.b, 1};
        }
        for (size_t i = 0; i < todo.size(); i++) entry[todo[i]] = ne[i];
        // (an entry that moved behind the next region's: that region's walk target changes too -- the walk of the region in
        // front is to this region's NEW entry only if that is where it arrived, which is how the entry was chosen)
        for (size_t i = 0; i < todo.size(); i++) walk(todo[i]);
    }
    // delivered: every live region's entry is where the walk in front arrived, up to the region where the input ended
    bool ok = true;
    uint64_t n_rsi = 0, last = 0;
    for (uint64_t r = 0; r < nreg && ok; r++) {
        if (!entry[r].live) continue;
        if (r && differs(r)) {
            ok = false;
            break;
        }
        last = r;
        if (ex[r].st) break;
    }
    stats[0] = nreg;
    stats[1] = kept;
    stats[2] = mism0;
    stats[3] = busy_passes;
    stats[4] = 0;
    if (!ok || ex[last].st != 1u) return 0;
    // fill
    uint64_t idx = 0;
    RgState x{};
    for (uint64_t r = 0; r <= last; r++) {
        if (!entry[r].live) continue;
        HostRing hr(s, c);
        auto &ps = hr.ps;
        x = RgState{entry[r].pos, entry[r].b, 0};
        const uint64_t nl = next_live(r);
        rg_walk(ps, c, x, nl < nreg && r != last ? entry[nl].pos : ~0ull, ~0ull,
                [&](uint64_t pos) {
                    if (idx < cap) rsi_off[idx] = pos;
                    idx++;
                    return true;
                },
                [](uint32_t, uint64_t) {});
    }
    n_rsi = idx;
    stats[4] = 1;
    stats[5] = n_rsi ? n_rsi - 1 : 0;       // complete RSIs: the last start met began one that is not (or nothing at all)
    stats[6] = x.b;
    stats[7] = x.pos;
    return 0;
}

}  // extern "C"
