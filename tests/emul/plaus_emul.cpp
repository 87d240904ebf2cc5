// plaus_emul.cpp -- entries by plausibility (libaec_amd/csrc/aec_idx.hip: k_lock_guess_p) restated
// on the CPU, chain by chain: the anchor by scoring chains from every bit of a stretch, the walk that looks one coded
// data set ahead with and without a reference sample, the confirmation by two chains, the entry from the count of blocks
// between the region's start and the RSI start found.  A MODEL of the kernel (its lanes and halves of a wavefront are
// loops here), with its constants; what it pins is the quality of the guesses on the reference's sample file -- how many
// entries are right BEFORE the exact machinery behind the guesses checks and repairs them.
// (test infrastructure; built by tests/test_plaus_emul.py)
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "../../libaec_amd/csrc/aec_cfg.h"
#include "../../libaec_amd/csrc/aec_trunk.h"

using namespace aec;

namespace {

constexpr uint32_t kSteps = 16, kAccept = 8, kConf = 8, kConfOk = 6, kLost = 3;    // (kLp* of aec_idx.hip)

struct Model {
    Cfg c;
    TrStream s;
    uint32_t id(uint64_t q) const { return (uint32_t)(tr_peek64(s, q) >> (64u - c.id_len)); }
    static bool near1(uint32_t a, uint32_t b) { return (a > b ? a - b : b - a) <= 1u; }
    static bool near2(uint32_t a, uint32_t b) { return (a > b ? a - b : b - a) <= 2u; }
    uint32_t step(uint64_t q, uint32_t ref, uint32_t &nz) const { return tr_cds(s, c, q, ref, nz); }

    uint64_t find_anchor(uint64_t from, uint32_t maxbits) const
    {
        uint32_t best = 0;
        uint64_t at = 0;
        for (uint64_t q0 = from; q0 < from + maxbits + 64; q0++) {
            uint64_t q = q0, anchor = 0;
            uint32_t sc = 0, prev = 0;
            bool ok = true;
            for (uint32_t i = 0; i < kSteps && ok; i++) {
                uint32_t nz;
                const uint32_t len = step(q, 0, nz);
                ok = len != 0;
                const uint32_t o = id(q);
                if (i && near1(o, prev)) sc++;
                prev = o;
                if (i == 3) anchor = q;
                q += len;
            }
            if (ok && sc > best) {
                best = sc;
                at = anchor;
            }
        }
        return best >= kAccept ? at : 0;
    }
    // 1: an RSI starts at q0; 2: the walk has lost the true chain; 0: on with the plain chain
    uint32_t confirm(uint64_t q0, bool with_ref) const
    {
        uint32_t sc[2] = {0, 0}, idp[2] = {id(q0), id(q0)}, so = 0, sr = 0;
        uint64_t v[2] = {q0, q0};
        bool ok[2] = {true, true};
        for (uint32_t k = 0; k < kConf; k++) {
            for (int h = 0; h < 2; h++) {
                uint32_t nz, len = step(v[h], k == 0 ? h : 0, nz);
                ok[h] = ok[h] && len;
                if (!ok[h]) continue;
                uint32_t idk = id(v[h] + len);
                if (k && !near2(idk, idp[h])) {       // (the chain may pass an RSI start)
                    uint32_t nz1;
                    const uint32_t l1 = step(v[h], 1, nz1);
                    if (l1 && near2(id(v[h] + l1), idp[h])) {
                        len = l1;
                        idk = id(v[h] + l1);
                    }
                }
                v[h] += len;
                sc[h] += near2(idk, idp[h]);
                idp[h] = idk;
            }
            so = ok[0] ? sc[0] : 0;
            sr = with_ref && ok[1] ? sc[1] : 0;
            const uint32_t done = k + 1;
            if (so >= sr + 2 && so >= 2) return 0;
            if (done - sr > kConf - kConfOk && (so >= kLost || done - so > kConf - kLost)) break;
        }
        if (sr >= kConfOk && sr > so) return 1;
        return so < kLost ? 2 : 0;
    }
};

}  // namespace

// stats: [0] regions, [1] entries right, [2] no guess, [3] wrong guess
extern "C" int emul_plaus(const uint32_t *prm, const uint8_t *stream, size_t nbytes, uint32_t regions_per_rsi, uint64_t *stats)
{
    Model m;
    if (make_cfg(prm[0], prm[1], prm[2], prm[3], 0, false, &m.c) != RC_OK) return -1;
    std::vector<uint8_t> buf(nbytes + 64, 0);
    memcpy(buf.data(), stream, nbytes);
    m.s = TrStream{reinterpret_cast<const uint32_t *>(buf.data()), (uint64_t)(nbytes + 3) / 4, (uint64_t)nbytes * 8};
    const Cfg &c = m.c;
    // the truth: every boundary and the count of blocks there, by the serial walk
    std::map<uint64_t, uint32_t> truth;
    uint64_t nstarts = 0;
    {
        uint64_t pos = 0;
        uint32_t b = 0;
        for (;;) {
            uint32_t nz;
            const uint32_t len = tr_cds(m.s, c, pos, (b == 0 && (c.flags & F_PREPROCESS)) ? 1u : 0u, nz);
            if (!len) break;
            truth[pos] = b;
            nstarts += b == 0;
            const uint32_t nb = tr_blocks(c, nz, b);
            if (!nb) break;
            pos += len;
            b += nb;
            if (b >= c.rsi) b = 0;
        }
    }
    if (!nstarts) return -2;
    const uint64_t total = (uint64_t)nbytes * 8, hint = total / nstarts;
    const uint32_t maxbits = c.id_len + 1 + c.bps + c.bs * c.bps;
    const uint64_t stride = hint / (regions_per_rsi ? regions_per_rsi : 1), back = 5 * maxbits + 128;
    memset(stats, 0, 4 * sizeof(uint64_t));
    for (uint64_t rstart = hint; rstart + hint < total; rstart += stride) {
        uint64_t q = m.find_anchor(rstart > back ? rstart - back : 0, maxbits);
        uint64_t cross = 0, S = 0;
        uint32_t steps = 0, nb_since = 0, reanch = 0;
        bool crossed = false, found = false, cnt_ok = true;
        while (q && steps++ < 2 * c.rsi + 64) {
            if (!crossed && q >= rstart) {
                crossed = true;
                cross = q;
                nb_since = 0;
            }
            const uint32_t o = m.id(q);
            uint32_t nz0, nz1;
            const uint32_t l0 = m.step(q, 0, nz0), l1 = m.step(q, 1, nz1);
            if (!l0) break;
            const bool nf = Model::near1(m.id(q + l0), o), ng = l1 && Model::near1(m.id(q + l1), o);
            if (!(nf && !ng)) {
                const uint32_t vd = m.confirm(q, l1 != 0);
                if (vd == 1) {
                    found = true;
                    S = q;
                    break;
                }
                if (vd == 2) {
                    if (crossed || ++reanch > 2) break;
                    q = m.find_anchor(q, maxbits);
                    continue;
                }
            }
            q += l0;
            if (crossed) {
                if (nz0 == 5) cnt_ok = false;
                nb_since += nz0 ? (nz0 > 5 ? nz0 - 1 : nz0) : 1;
            }
        }
        bool have = false;
        uint64_t e_pos = 0;
        uint32_t e_b = 0;
        if (found && crossed && cnt_ok) {
            have = true;
            e_pos = cross;
            e_b = (c.rsi - nb_since % c.rsi) % c.rsi;
        } else if (found && !crossed) {          // exact from the RSI start on
            uint64_t p = S;
            uint32_t b = 0;
            while (p < rstart) {
                uint32_t nz;
                const uint32_t len = tr_cds(m.s, c, p, b == 0, nz);
                if (!len) break;
                const uint32_t nb = tr_blocks(c, nz, b);
                p += len;
                b += nb;
                if (b >= c.rsi) b = 0;
            }
            have = true;
            e_pos = p;
            e_b = b;
        }
        auto it = truth.lower_bound(rstart);
        const bool right = have && it != truth.end() && it->first == e_pos && it->second == e_b;
        stats[0]++;
        stats[1] += right;
        stats[2] += !have;
        stats[3] += have && !right;
    }
    return 0;
}
