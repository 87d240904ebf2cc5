// emul.cpp -- CPU harness for libaec_amd/csrc/aec_lane.h (TEST INFRASTRUCTURE).
// Runs the per-lane device functions lane by lane with the wave-level glue (ballots,
// prefix sums, k-clamp scan, word assembly) written as plain loops, so the arithmetic and the
// parallel reformulation can be checked against the oracle without a GPU.  Not linked into
// the product library.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../libaec_amd/csrc/aec_lane.h"
#include "../../libaec_amd/csrc/aec_cfg.h"

using namespace aec;

// word source over plain memory for WinReader (the device uses an LDS ring instead)
struct MemSrc {
    const uint32_t *w;
    uint64_t n, base;
    uint32_t word(uint32_t i) const { const uint64_t idx = base + i; return idx < n ? bswap32(w[idx]) : 0u; }
    void word2(uint32_t i, uint32_t &w0, uint32_t &w1) const { w0 = word(i); w1 = word(i + 1); }
    void word3(uint32_t i, uint32_t &w0, uint32_t &w1, uint32_t &w2) const { w0 = word(i); w1 = word(i + 1); w2 = word(i + 2); }
    bool starved() const { return false; }
};

struct VecSink {
    std::vector<uint32_t> &w;
    void or_word(uint32_t i, uint32_t v) { if (i >= w.size()) w.resize(i + 1, 0); w[i] |= v; }
};

template <bool WIDE>
static int encode_t(const Cfg &c, const uint8_t *in, uint8_t *out, size_t cap, uint32_t start_bit,
                    uint32_t k_in, uint64_t *total_bits, uint32_t *k_out, uint32_t *meta_out,
                    uint64_t *rsi_off)
{
    const bool pp = c.flags & F_PREPROCESS, msb = c.flags & F_MSB;
    const uint32_t mask = low_mask32(c.bps);
    uint64_t bitpos = start_bit;
    uint32_t kcur = k_in;
    memset(out, 0, cap);
    std::vector<uint32_t> d(64 * 64);
    for (uint64_t sg = 0; sg < c.total_segs; sg++) {
        const uint64_t r = sg / c.segs_per_rsi;
        const uint32_t s = (uint32_t)(sg % c.segs_per_rsi);
        const uint32_t b0 = s * 64;
        uint64_t nb_rsi = c.total_blocks - r * c.rsi;
        if (nb_rsi > c.rsi) nb_rsi = c.rsi;
        const uint32_t nv = (uint32_t)((nb_rsi - b0) < 64 ? (nb_rsi - b0) : 64);
        const uint64_t samp0 = (r * c.rsi + b0) * (uint64_t)c.bs;
        auto raw = [&](uint64_t i) {
            if (i >= c.total_samples) i = c.total_samples - 1;
            return load_sample_bytes(in + i * c.bytes, c.bytes, msb);
        };
        if (s == 0 && rsi_off) rsi_off[r] = bitpos;
        const uint32_t ref_sample = raw(r * c.rsi * (uint64_t)c.bs) & mask;
        // phase A: preprocess
        for (uint32_t i = 0; i < nv * c.bs; i++) {
            const uint64_t gi = samp0 + i;
            uint32_t v;
            if (!pp) v = raw(gi);
            else if (b0 == 0 && i == 0) v = 0;
            else v = pp_any(raw(gi - 1), raw(gi), c);
            d[i] = v;
        }
        // phase B: per lane analysis
        uint32_t meta[64]; uint64_t zmask = 0;
        BlockChoice ch[64];
        for (uint32_t l = 0; l < nv; l++) {
            bool z = true;
            for (uint32_t i = 0; i < c.bs; i++) if (d[l * c.bs + i]) z = false;
            if (z) zmask |= 1ull << l;
        }
        KClamp segc = kclamp_identity();
        uint32_t kprev[64];
        for (uint32_t l = 0; l < nv; l++) {
            const uint32_t ref = (pp && b0 == 0 && l == 0) ? 1 : 0;
            kprev[l] = kclamp_apply(segc, kcur);
            if ((zmask >> l) & 1) {
                uint32_t fs;
                const uint32_t run = zero_run_at(zmask, l, nv, fs);
                if (run == 0) meta[l] = meta_pack(0, OPT_ZCONT, 0, 0);
                else meta[l] = meta_pack(c.id_len + 1 + ref * c.bps + fs + 1, OPT_ZERO, fs, 0);
            } else {
                ch[l] = choose_option<0, WIDE>(&d[l * c.bs], c, ref);
                meta[l] = meta_pack(ch[l].bits, ch[l].opt, ch[l].klo, ch[l].khi);
                if (c.id_len > 1) segc = kclamp_then(segc, KClamp{ch[l].klo, ch[l].khi});
            }
            if (meta_out) meta_out[(r * c.rsi + b0) + l] = meta[l];
        }
        // pack
        uint32_t lead = (uint32_t)(bitpos & 31);
        std::vector<uint32_t> words;
        VecSink sink{words};
        uint32_t off = lead;
        for (uint32_t l = 0; l < nv; l++) {
            const uint32_t ref = (pp && b0 == 0 && l == 0) ? 1 : 0;
            const uint32_t opt = meta_opt(meta[l]);
            if (opt == OPT_ZCONT) continue;
            uint32_t karg = opt == OPT_ZERO ? meta_a(meta[l])
                           : kclamp_apply(KClamp{meta_a(meta[l]), meta_b(meta[l])}, kprev[l]);
            BitWriter<VecSink> w(sink, off);
            uint32_t ubits = 0, fbits = 0;
            const bool small = (c.bs == 8 || c.bs == 16) &&
                               small_eligible(c, c.bs, opt, karg, ref, meta_len(meta[l]), ubits, fbits);
            if (small && c.bs == 8) emit_small<8>(w, &d[l * c.bs], c, opt, karg, ref, ref_sample, ubits, fbits, true);
            else if (small) emit_small<16>(w, &d[l * c.bs], c, opt, karg, ref, ref_sample, ubits, fbits, true);
            else if (opt == OPT_SPLIT && c.bs == 32) emit_split_groups<32>(w, &d[l * c.bs], c, karg, ref, ref_sample, true);
            else if (opt == OPT_SPLIT && c.bs == 64) emit_split_groups<64>(w, &d[l * c.bs], c, karg, ref, ref_sample, true);
            else emit_block<0>(w, &d[l * c.bs], c, opt, karg, ref, ref_sample);
            off += meta_len(meta[l]);
        }
        kcur = kclamp_apply(segc, kcur);
        const uint64_t gw = bitpos >> 5;
        for (size_t w = 0; w < words.size(); w++) {
            const uint32_t v = bswap32(words[w]);
            const size_t byte = (gw + w) * 4;
            for (int b = 0; b < 4; b++)
                if (byte + b < cap) out[byte + b] |= (uint8_t)(v >> (8 * b));
        }
        bitpos += off - lead;
    }
    if (rsi_off) rsi_off[c.rsi_count] = bitpos;
    *total_bits = bitpos - start_bit;
    *k_out = kcur;
    return 0;
}

extern "C" int emul_encode(const uint32_t *p /*bps,bs,rsi,flags*/, const uint8_t *in, size_t in_len,
                           uint8_t *out, size_t cap, uint32_t start_bit, uint32_t k_in,
                           uint64_t *total_bits, uint32_t *k_out, uint32_t *meta_out, uint64_t *rsi_off)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], in_len, true, &c);
    if (rc) return rc;
    return c.bps > 16 ? encode_t<true>(c, in, out, cap, start_bit, k_in, total_bits, k_out, meta_out, rsi_off)
                      : encode_t<false>(c, in, out, cap, start_bit, k_in, total_bits, k_out, meta_out, rsi_off);
}

// phase-structured block decoder (aec_lane.h decode_block), templated block sizes only
template <int BS>
static int decode_fast(const Cfg &c, const std::vector<uint32_t> &words, size_t in_len, const uint64_t *rsi_off,
                       uint64_t nrsi, uint64_t total_blocks, uint8_t *out, size_t cap)
{
    const bool pp = c.flags & F_PREPROCESS, msb = c.flags & F_MSB;
    uint32_t d[BS];
    for (uint64_t r = 0; r < nrsi; r++) {
        const uint64_t a0 = (rsi_off[r] >> 5) & ~3ull;
        MemSrc src{words.data(), words.size(), a0};
        uint32_t p = (uint32_t)(rsi_off[r] - a0 * 32);
        const uint32_t end_p = (uint32_t)((uint64_t)in_len * 8 - a0 * 32);
        uint64_t nb = total_blocks - r * c.rsi;
        if (nb > c.rsi) nb = c.rsi;
        uint32_t x = 0, zrun = 0;
        uint64_t o = r * c.rsi * (uint64_t)c.bs;
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t ref = (pp && b == 0) ? 1 : 0;
            uint32_t nz = 0;
            const bool parse = zrun == 0;
            const uint32_t st = decode_block<BS>(src, p, end_p, d, c, ref, b, parse, nz);
            if (parse && st != DEC_OK) return (int)st;
            if (parse && nz) zrun = nz;
            for (uint32_t j = 0; j < (uint32_t)BS; j++) {
                uint32_t v;
                if (!pp) v = d[j];
                else if (ref && parse && j == 0) v = x = (c.flags & F_SIGNED) ? sign_extend(d[0], c.bps) : d[0];
                else v = x = (c.flags & F_SIGNED) ? unpp_signed(x, d[j], c.xmax) : unpp_unsigned(x, d[j], c.xmax);
                if ((o + j + 1) * c.bytes > cap) return -100;
                uint8_t *q = out + (o + j) * c.bytes;
                for (uint32_t t = 0; t < c.bytes; t++)
                    q[t] = (uint8_t)(v >> (8 * (msb ? c.bytes - 1 - t : t)));
            }
            o += BS;
            if (zrun) zrun--;
        }
    }
    return 0;
}

// decode every RSI from its bit offset; returns status, writes whole blocks
extern "C" int emul_decode(const uint32_t *p, const uint8_t *in, size_t in_len, const uint64_t *rsi_off,
                           uint64_t nrsi, uint64_t total_blocks, uint8_t *out, size_t cap)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const bool pp = c.flags & F_PREPROCESS, msb = c.flags & F_MSB;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    switch (c.bs) {
    case 8: return decode_fast<8>(c, words, in_len, rsi_off, nrsi, total_blocks, out, cap);
    case 16: return decode_fast<16>(c, words, in_len, rsi_off, nrsi, total_blocks, out, cap);
    case 32: return decode_fast<32>(c, words, in_len, rsi_off, nrsi, total_blocks, out, cap);
    case 64: return decode_fast<64>(c, words, in_len, rsi_off, nrsi, total_blocks, out, cap);
    default: break;
    }
    uint32_t d[64];
    for (uint64_t r = 0; r < nrsi; r++) {
        WinReader<MemSrc> br;
        const uint64_t a0 = (rsi_off[r] >> 5) & ~3ull;
        br.init(MemSrc{words.data(), words.size(), a0}, a0 * 32, (uint64_t)in_len * 8,
                (uint32_t)(rsi_off[r] - a0 * 32));
        uint64_t nb = total_blocks - r * c.rsi;
        if (nb > c.rsi) nb = c.rsi;
        uint32_t x = 0;
        uint64_t o = r * c.rsi * (uint64_t)c.bs;
        uint32_t b = 0;
        while (b < nb) {
            const uint32_t ref = (pp && b == 0) ? 1 : 0;
            uint32_t nz;
            const uint32_t st = parse_cds<0>(br, d, c, ref, b, nz);
            if (st != DEC_OK) return (int)st;
            uint32_t nblk = nz ? nz : 1;
            // a rest-of-segment run in a short final RSI is closed by the end of the data, not by
            // the nominal RSI length: never produce more blocks than the caller expects
            if (nblk > nb - b) nblk = (uint32_t)(nb - b);
            for (uint32_t j = 0; j < nblk * c.bs; j++) {
                uint32_t v;
                const uint32_t dv = nz ? ((j == 0 && ref) ? d[0] : 0) : d[j];
                if (!pp) v = dv;
                else if (ref && j == 0) v = x = (c.flags & F_SIGNED) ? sign_extend(dv, c.bps) : dv;
                else v = x = (c.flags & F_SIGNED) ? unpp_signed(x, dv, c.xmax) : unpp_unsigned(x, dv, c.xmax);
                uint8_t *q = out + (o + j) * c.bytes;
                if ((o + j + 1) * c.bytes > cap) return -100;
                for (uint32_t t = 0; t < c.bytes; t++)
                    q[t] = (uint8_t)(v >> (8 * (msb ? c.bytes - 1 - t : t)));
            }
            o += (uint64_t)nblk * c.bs;
            b += nblk;
        }
    }
    return 0;
}

// serial RSI index: walks the stream from bit `start_bit` (an RSI start) and records offsets
extern "C" int emul_index(const uint32_t *p, const uint8_t *in, size_t in_len, uint64_t start_bit,
                          uint64_t *rsi_off, uint64_t max_rsi, uint64_t *n_rsi, uint64_t *tail_blocks,
                          uint64_t *end_bit)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const bool pp = c.flags & F_PREPROCESS;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    BitReader br;
    br.init(words.data(), words.size(), (uint64_t)in_len * 8, start_bit);
    uint64_t r = 0; uint32_t b = 0; uint64_t good = start_bit;
    int status = DEC_OK;
    for (;;) {
        if (b == 0) { if (r >= max_rsi) break; rsi_off[r] = good; }
        uint32_t nblk;
        BitReader save = br;
        const uint32_t st = skip_cds(br, c, (pp && b == 0) ? 1 : 0, b, nblk);
        if (st != DEC_OK) { br = save; status = st == DEC_NEED_INPUT ? 0 : (int)st; break; }
        good = br.pos;
        b += nblk;
        if (b >= c.rsi) { b = 0; r++; }
    }
    *n_rsi = r; *tail_blocks = b; *end_bit = good;
    return status;
}

// ---- speculative RSI index (aec_spec.h), window by window as k_spec does it --------------------
#include <algorithm>
#include "../../libaec_amd/csrc/aec_spec.h"

// Fills T[p] (RSI length for a hypothetical RSI start at bit p, 0 = unresolved), Xb/Xc (chained
// hop out of the window core: bits, RSI count) for every bit position of the stream.
extern "C" int emul_spec(const uint32_t *p, const uint8_t *in, size_t in_len, uint32_t core, uint32_t look,
                         uint16_t *T, uint16_t *Xb, uint8_t *Xc)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const uint64_t end_bit = (uint64_t)in_len * 8;
    const uint32_t W = core + look, nw = W / 32;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    std::vector<uint32_t> win(nw + 2);
    std::vector<uint16_t> rank(nw + 1), sel(nw + 2), nxt(W), nxt4(W), nxt16(W), Tl(core);
    const bool pad = c.flags & F_PAD_RSI;
    for (uint64_t wstart = 0; wstart < end_bit; wstart += core) {
        for (uint32_t i = 0; i < nw + 2; i++) {
            const uint64_t idx = wstart / 32 + i;
            win[i] = idx < words.size() ? bswap32(words[idx]) : 0u;
        }
        rank[0] = 0;
        for (uint32_t i = 0; i < nw; i++) rank[i + 1] = (uint16_t)(rank[i] + __builtin_popcount(win[i]));
        for (uint32_t i = 0; i < nw; i++) {
            const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
            if (32u * m + 1u <= hi && 32u * m + 1u > lo) sel[m] = (uint16_t)i;
        }
        SpecWin s{win.data(), rank.data(), sel.data(), nw,
                  (uint32_t)std::min<uint64_t>(W, end_bit - wstart)};
        for (uint32_t q = 0; q < W; q++) nxt[q] = q < s.limit ? spec_nxt_entry(s, c, q) : 0;
        for (uint32_t q = 0; q < W; q++) nxt4[q] = spec_hop4(nxt.data(), c, s.limit, q);
        for (uint32_t q = 0; q < W; q++) nxt16[q] = spec_hop16(nxt4.data(), s.limit, q);
        const uint32_t variant = (uint32_t)(wstart / core) % 3u;   // exercise the walk with and without hop tables
        for (uint32_t q = 0; q < core && wstart + q < end_bit; q++) {
            const uint32_t t = spec_rsi(s, c, nxt.data(), variant ? nxt4.data() : nullptr,
                                        variant == 2 ? nxt16.data() : nullptr, q);
            Tl[q] = (uint16_t)(t <= 0xFFFFu ? t : 0);
            T[wstart + q] = Tl[q];
        }
        for (uint32_t q = 0; q < core && wstart + q < end_bit; q++) {
            uint32_t pos = q, cnt = 0;
            while (pos < core && wstart + pos < end_bit && Tl[pos] && cnt < 255) {
                pos += Tl[pos];
                cnt++;
                if (pad) pos = (pos + 7u) & ~7u;
            }
            Xb[wstart + q] = (uint16_t)(pos - q);
            Xc[wstart + q] = (uint8_t)cnt;
        }
    }
    return 0;
}

// ---- sparse speculative index (aec_spec2.h), window by window as k_spec2 does it ------------------
#define AEC_S2_COUNT 1
#include <unordered_set>
static std::unordered_set<uint64_t> s2_noted;      // distinct (window, position) pairs parsed on demand
static uint64_t s2_note_window = 0;
#define S2_NOTE(pos) (s2_noted.insert((s2_note_window << 24) | (pos)))
static unsigned long long s2_hist_parse[256], s2_hist_table[256];
#define S2_HIST(parses, steps) (s2_hist_parse[(parses) < 255 ? (parses) : 255]++, s2_hist_table[(steps) < 255 ? (steps) : 255]++)
extern "C" void emul_s2_hist(unsigned long long *parse, unsigned long long *table)
{
    for (int i = 0; i < 256; i++) {
        parse[i] = s2_hist_parse[i];
        table[i] = s2_hist_table[i];
        s2_hist_parse[i] = s2_hist_table[i] = 0;
    }
}
#include "../../libaec_amd/csrc/aec_spec2.h"
// step statistics of the unit walks (hop16, hop4, table single, on-demand single, units)
extern "C" void emul_s2_counters(unsigned long long *out, int reset)
{
    for (int i = 0; i < 8; i++) {
        out[i] = aec::s2_counters[i];
        if (reset) aec::s2_counters[i] = 0;
    }
    out[7] = s2_noted.size();
    if (reset) s2_noted.clear();
}

// prm: core, lead, look, stride, burn, mode (0 = unit is the RSI, 1 = units are segments of 64 blocks),
// marking steps per chain (0 = unbounded), on-demand parses per unit walk (0 = unbounded).
// Output, DENSE for checking (the kernel writes the same values sparsely): marked[p] = 1 where bit p of
// the stream is a candidate inside its window's core; recs[p] = its record.
extern "C" int emul_spec2(const uint32_t *p, const uint8_t *in, size_t in_len, const uint32_t *prm,
                          uint64_t start_bit, uint8_t *marked, S2Rec *recs)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const uint32_t core = prm[0], lead = prm[1], look = prm[2], stride = prm[3], burn = prm[4], mode = prm[5];
    const uint32_t max_mark = prm[6] ? prm[6] : 0xFFFFFFFFu;   // marking steps per chain (0 = unbounded)
    const uint32_t budget = prm[7] ? prm[7] : 0xFFFFFFFFu;     // on-demand parses per unit walk (0 = unbounded)
    const uint64_t end_bit = (uint64_t)in_len * 8;
    const uint32_t W = lead + core + look, nw = W / 32;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    std::vector<uint32_t> win(nw + 2), marks(nw);
    std::vector<uint16_t> rank(nw + 1), sel(nw + 2), mpre(nw + 1);
    const uint64_t tab_lo = start_bit / core * core;
    for (uint64_t core_abs = tab_lo; core_abs < end_bit; core_abs += core) {
        const uint64_t wstart = core_abs >= lead ? core_abs - lead : 0;      // (multiple of 32)
        const uint32_t c0 = (uint32_t)(core_abs - wstart), c1 = c0 + core;
        s2_note_window = core_abs / core;
        for (uint32_t i = 0; i < nw + 2; i++) {
            const uint64_t idx = wstart / 32 + i;
            win[i] = idx < words.size() ? bswap32(words[idx]) : 0u;
        }
        rank[0] = 0;
        for (uint32_t i = 0; i < nw; i++) rank[i + 1] = (uint16_t)(rank[i] + __builtin_popcount(win[i]));
        for (uint32_t i = 0; i < nw; i++) {
            const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
            if (32u * m + 1u <= hi && 32u * m + 1u > lo) sel[m] = (uint16_t)i;
        }
        SpecWin s{win.data(), rank.data(), sel.data(), nw, (uint32_t)std::min<uint64_t>(W, end_bit - wstart)};
        // 1. sync chains
        std::fill(marks.begin(), marks.end(), 0u);
        auto mark = [&](uint32_t q) {
            const uint32_t bit = 1u << (31u - (q & 31u));
            const bool was = marks[q >> 5] & bit;
            marks[q >> 5] |= bit;
            return was;
        };
        if (start_bit >= wstart && start_bit - wstart < s.limit) mark((uint32_t)(start_bit - wstart));
        for (uint32_t q0 = 0; q0 < s.limit; q0 += stride) {
            uint32_t q = q0;
            bool ok = true;
            for (uint32_t k = 0; k < burn && ok; k++) {
                const uint32_t len = s2_chain_step(s, c, q);
                ok = len != 0;
                q += len;
            }
            for (uint32_t steps = 0; ok && q < s.limit && steps < max_mark; steps++) {
                if (mark(q)) break;
                const uint32_t len = s2_chain_step(s, c, q);
                ok = len != 0;
                q += len;
            }
        }
        // 2. candidate tables
        mpre[0] = 0;
        for (uint32_t i = 0; i < nw; i++) mpre[i + 1] = (uint16_t)(mpre[i] + __builtin_popcount(marks[i]));
        const uint32_t ncand = mpre[nw];
        std::vector<uint16_t> cpos(ncand + 1), cnxt(ncand + 1), csucc(ncand + 1), chop4(ncand + 1), chop16(ncand + 1);
        for (uint32_t q = 0, i = 0; q < W; q++)
            if (s2_marked(marks.data(), q)) cpos[i++] = (uint16_t)q;
        S2Win w{s, marks.data(), mpre.data(), cnxt.data(), csucc.data(), chop4.data(), chop16.data(), cpos.data(), ncand};
        for (uint32_t i = 0; i < ncand; i++) cnxt[i] = cpos[i] < s.limit ? spec_nxt_entry(s, c, cpos[i]) : 0;
        for (uint32_t i = 0; i < ncand; i++) csucc[i] = s2_succ(w, i);
        for (uint32_t i = 0; i < ncand; i++) chop4[i] = s2_hop4(w, c, i);
        for (uint32_t i = 0; i < ncand; i++) chop16[i] = s2_hop16(w, i);
        // 3. units and chains for the candidates of the core
        std::vector<uint32_t> ua(ncand + 1, 0), um(ncand + 1, 0);
        for (uint32_t i = 0; i < ncand; i++) {
            const uint32_t q = cpos[i];
            if (q < c0 || q >= c1 || q >= s.limit) continue;
            if (mode == 0) {
                ua[i] = s2_unit(w, c, q, 0, c.rsi, budget);
            } else {
                ua[i] = s2_unit(w, c, q, 0, 64, budget);
                um[i] = s2_unit(w, c, q, 64, 128, budget);
            }
        }
        auto chain = [&](const std::vector<uint32_t> &u, uint32_t i) -> uint32_t {
            uint32_t pos = cpos[i], cnt = 0;
            while (pos < c1 && pos < s.limit && cnt < 255u) {
                const uint32_t j = s2_index(w, pos);
                if (j == kS2NoIndex || !u[j]) break;
                pos += u[j];
                cnt++;
            }
            return cnt ? ((cnt << 24) | (pos - cpos[i])) : 0u;
        };
        for (uint32_t i = 0; i < ncand; i++) {
            const uint32_t q = cpos[i];
            if (q < c0 || q >= c1 || q >= s.limit) continue;
            const uint64_t abs = wstart + q;
            marked[abs] = 1;
            recs[abs] = S2Rec{ua[i], chain(ua, i), um[i], mode ? chain(um, i) : 0u};
        }
    }
    return 0;
}

// spec_cds_fast against spec_cds at EVERY bit position of the stream's windows (with and without a reference
// sample): wherever the fast parse answers, the answers must be the same.  out[0] = positions compared,
// out[1] = unresolved by the fast parse, out[2] = mismatches, out[3] = first mismatching absolute bit.
extern "C" int emul_cds_fast_check(const uint32_t *p, const uint8_t *in, size_t in_len, uint32_t W, uint64_t *out)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const uint64_t end_bit = (uint64_t)in_len * 8;
    const uint32_t nw = W / 32;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    std::vector<uint32_t> win(nw + 4);
    std::vector<uint16_t> rank(nw + 1), sel(nw + 2);
    out[0] = out[1] = out[2] = 0;
    out[3] = ~0ull;
    for (uint64_t wstart = 0; wstart < end_bit; wstart += W / 2) {
        for (uint32_t i = 0; i < nw + 4; i++) {
            const uint64_t idx = wstart / 32 + i;
            win[i] = idx < words.size() ? bswap32(words[idx]) : 0u;
        }
        rank[0] = 0;
        for (uint32_t i = 0; i < nw; i++) rank[i + 1] = (uint16_t)(rank[i] + __builtin_popcount(win[i]));
        for (uint32_t i = 0; i < nw; i++) {
            const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
            if (32u * m + 1u <= hi && 32u * m + 1u > lo) sel[m] = (uint16_t)i;
        }
        SpecWin s{win.data(), rank.data(), sel.data(), nw, (uint32_t)std::min<uint64_t>(W, end_bit - wstart)};
        for (uint32_t q = 0; q < s.limit; q++) {
            for (uint32_t ref = 0; ref < 2; ref++) {
                uint32_t run_s, run_f;
                const uint32_t ls = spec_cds(s, c, q, ref, run_s);
                const uint32_t lf = ref ? spec_cds_fast<1>(s.win, s.limit, c, q, run_f)
                                        : spec_cds_fast<0>(s.win, s.limit, c, q, run_f);
                uint32_t run_g;
                const uint32_t lg = spec_cds_fast<2>(s.win, s.limit, c, q, run_g, ref);
                if (lg != lf || run_g != run_f) {
                    if (!out[2]) out[3] = wstart + q;
                    out[2]++;
                }
                out[0]++;
                if (lf == kSpecUnresolved) {
                    out[1]++;
                } else if (lf != ls || (lf && run_f != run_s)) {
                    if (!out[2]) out[3] = wstart + q;
                    out[2]++;
                }
            }
        }
    }
    return 0;
}

// statistics helper for tests: for start positions q0 = first, first + step, ... the number of CDS
// parses (chain steps, no reference sample) until the chain from q0 lands on a TRUE boundary
// (truth[] = 1 at true coded-data-set boundaries), capped at max_steps; out[i] = steps or 0xFFFF.
extern "C" int emul_sync_steps(const uint32_t *p, const uint8_t *in, size_t in_len, const uint8_t *truth,
                               uint64_t first, uint64_t step, uint32_t n, uint32_t max_steps, uint16_t *out)
{
    Cfg c;
    int rc = make_cfg(p[0], p[1], p[2], p[3], 0, false, &c);
    if (rc) return rc;
    const uint32_t W = 65536, nw = W / 32;
    std::vector<uint32_t> words((in_len + 3) / 4 + 1, 0);
    memcpy(words.data(), in, in_len);
    std::vector<uint32_t> win(nw + 2);
    std::vector<uint16_t> rank(nw + 1), sel(nw + 2);
    const uint64_t end_bit = (uint64_t)in_len * 8;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t q0 = first + (uint64_t)i * step;
        const uint64_t wstart = q0 & ~31ull;
        for (uint32_t k = 0; k < nw + 2; k++) {
            const uint64_t idx = wstart / 32 + k;
            win[k] = idx < words.size() ? bswap32(words[idx]) : 0u;
        }
        rank[0] = 0;
        for (uint32_t k = 0; k < nw; k++) rank[k + 1] = (uint16_t)(rank[k] + __builtin_popcount(win[k]));
        for (uint32_t k = 0; k < nw; k++) {
            const uint32_t lo = rank[k], hi = rank[k + 1], m = (lo + 31u) >> 5;
            if (32u * m + 1u <= hi && 32u * m + 1u > lo) sel[m] = (uint16_t)k;
        }
        SpecWin s{win.data(), rank.data(), sel.data(), nw, (uint32_t)std::min<uint64_t>(W, end_bit - wstart)};
        uint32_t q = (uint32_t)(q0 - wstart), steps = 0;
        out[i] = 0xFFFF;
        while (steps < max_steps && q < s.limit) {
            if (truth[wstart + q]) { out[i] = (uint16_t)steps; break; }
            const uint32_t len = s2_chain_step(s, c, q);
            if (!len) break;
            q += len;
            steps++;
        }
    }
    return 0;
}

// second-extension code -> (sum, second) for every code value (tests the closed form in aec_lane.h)
extern "C" int emul_se_lookup(uint32_t m, uint32_t *sum, uint32_t *second)
{
    return aec::se_lookup(m, *sum, *second) ? 1 : 0;
}
