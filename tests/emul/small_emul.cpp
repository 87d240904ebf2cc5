// small_emul.cpp -- the every-bit index scheme (libaec_amd/csrc/aec_small.h; aec_idx.hip: launch_index_small) on the
// CPU: the per-bit functions the kernels are loops over, run bit by bit over a whole stream, the doubling in base 4
// round by round as the launches do it, and the RSI starts compared with those the oracle's encoder reports.
// (test infrastructure; built by tests/test_small_emul.py)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../libaec_amd/csrc/aec_cfg.h"
#include "../../libaec_amd/csrc/aec_small.h"

using namespace aec;

extern "C" int emul_small(const uint32_t *prm, const uint8_t *stream, size_t nbytes, const uint64_t *want, uint64_t nwant,
                          uint32_t use_hops, uint64_t *stats)
{
    Cfg c;
    if (make_cfg(prm[0], prm[1], prm[2], prm[3], 0, false, &c) != RC_OK) return -1;
    std::vector<uint8_t> buf(nbytes + 64, 0);
    memcpy(buf.data(), stream, nbytes);
    const TrStream s{reinterpret_cast<const uint32_t *>(buf.data()), (uint64_t)(nbytes + 3) / 4, (uint64_t)nbytes * 8};
    const uint32_t nbits = (uint32_t)(nbytes * 8);
    std::vector<uint16_t> e0(nbits + 1, 0), e1(nbits + 1, 0);
    for (uint32_t q = 0; q < nbits; q++) sm_parse(s, c, q, e0[q], e1[q]);
    auto r0 = [&](uint32_t at) { return (uint32_t)e0[at]; };
    auto r1 = [&](uint32_t at) { return (uint32_t)e1[at]; };
    std::vector<uint32_t> hop(nbits + 1, 0), ja(nbits + 1, kSmNone), jb(nbits + 1, kSmNone);
    if (use_hops)
        for (uint32_t q = 0; q <= nbits; q++) hop[q] = sm_hop(c, r0, q, nbits);
    auto rh = [&](uint32_t at) { return hop[at]; };
    std::vector<uint32_t> hop2(nbits + 1, 0);
    if (use_hops > 1)
        for (uint32_t q = 0; q <= nbits; q++) hop2[q] = sm_hop2(rh, q, nbits);
    auto rh2 = [&](uint32_t at) { return hop2[at]; };
    uint64_t differ = 0;
    for (uint32_t q = 0; q <= nbits; q++) {
        ja[q] = sm_rsi(c, r0, r1, rh, use_hops != 0, rh2, use_hops > 1, q, nbits);
        // (the walk through the hops must be the walk without them)
        if (use_hops && ja[q] != sm_rsi(c, r0, r1, rh, false, rh2, false, q, nbits)) differ++;
    }
    uint32_t levels = 0;
    while ((1ull << (2u * levels)) < nwant + 2) levels++;
    const uint32_t scap = 1u << (2u * levels);
    std::vector<uint32_t> sidx(scap, kSmNone);
    sidx[0] = 0;
    uint32_t *j = ja.data(), *jn = jb.data();
    for (uint32_t k = 0; k < levels; k++) {
        const uint32_t quarter = 1u << (2u * k);
        // (a launch reads the starts of the rounds before it and writes others: order within a round does not matter)
        for (uint32_t i = quarter; i-- > 0;) sm_double_starts(j, sidx.data(), i, quarter, scap);
        if (k + 1 < levels) {
            for (uint32_t q = 0; q <= nbits; q++) jn[q] = sm_double_table(j, q);
            uint32_t *t = j;
            j = jn;
            jn = t;
        }
    }
    uint64_t m = 0;
    while (m < scap && sidx[m] != kSmNone) m++;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < nwant && i < m; i++) bad += sidx[i] != want[i];
    stats[0] = m;          // RSI starts on the chain
    stats[1] = bad;        // ... that are not the encoder's
    stats[2] = differ;     // positions where hops change the walk
    stats[3] = levels;
    return 0;
}
