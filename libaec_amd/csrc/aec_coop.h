// aec_coop.h -- one WAVEFRONT parses one chain of coded data sets (device only; aec_idx.hip).
//
// A chain of coded data sets is serial: where one ends is where the next starts (reference src/decode.c:402-421).
// A single lane that follows such a chain through device memory pays a memory round trip and ~280 dependent
// vector instructions per coded data set (aec_trunk.h: tr_cds), 2-3 us; the kernels that are bound by the LONGEST
// chain they hold -- the regions of the trunk, the hypothesis walks that were handed on, the walks that look for
// segment starts -- are bound by that.  Here the 64 lanes of a wavefront hold 64 consecutive stream words (2048
// bits) in a register each, served from a window of the stream in LDS that they refill together with coalesced
// 16-byte loads; the header of a coded data set is two lane reads, the end of its unary part a masked popcount per
// lane, one DPP prefix sum, a ballot and a rank select inside one word: ~70 wave instructions and no memory round
// trip per coded data set.  The result is tr_cds' (length in bits, zero-run code), bit for bit: whatever does not
// end inside the register window is handed to tr_cds itself.
//
// Workgroups of ONE wavefront (the window refill synchronises with __syncthreads()).
#pragma once

#include "aec_trunk.h"

namespace aec {

template <uint32_t WINW>            // words of the LDS window (a multiple of 256)
struct CoopCds {
    TrStream s;
    uint32_t *win;                  // LDS, WINW words in host order
    uint64_t base;                  // stream word index of win[0]
    uint64_t wbase;                 // stream word held by lane 0
    uint32_t W;                     // this lane's word of the register window
    uint32_t lane, maxbits, idmax;
    bool loaded;
    // optional: the trunk marks of the same stretch (aec_trunk.h TrTables::bitmap), so that "does the walk stand on the
    // trunk" is an LDS read as well
    uint32_t *bmw;                  // LDS, WINW words, or null
    const uint32_t *bitmap;
    uint64_t bm_lo, bm_words;       // bit position of bitmap bit 0, words of the bitmap

    __device__ __forceinline__ void init(const TrStream &stream, const Cfg &c, uint32_t *lds_window)
    {
        s = stream;
        win = lds_window;
        lane = threadIdx.x & 63u;
        maxbits = c.id_len + 1u + c.bps + c.bs * c.bps;
        idmax = (1u << c.id_len) - 1u;
        base = ~0ull;
        wbase = 0;
        W = 0;
        loaded = false;
        bmw = nullptr;
        bitmap = nullptr;
        bm_lo = 0;
        bm_words = 0;
    }
    __device__ __forceinline__ void with_marks(uint32_t *lds_marks, const TrGeom &g, const TrTables &t)
    {
        bmw = lds_marks;
        bitmap = t.bitmap;
        bm_lo = g.lo;
        bm_words = (uint64_t)g.nwin * (g.L / 32u);
    }
    // does the register-window parse serve this parameter set at all (a coded data set of the encoder's kind plus the
    // look-ahead of the searches inside 2048 bits)?
    __device__ __forceinline__ bool usable() const { return maxbits + 128u <= 2048u; }

    __device__ __forceinline__ void refill(uint64_t from_word)
    {
        base = from_word & ~3ull;
        __syncthreads();
        for (uint32_t i = lane * 4u; i < WINW; i += 64u * 4u) {
            const uint64_t idx = base + i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx + 4u <= s.nwords) {
                v = *reinterpret_cast<const uint4 *>(s.words + idx);
            } else {
                if (idx < s.nwords) v.x = s.words[idx];
                if (idx + 1u < s.nwords) v.y = s.words[idx + 1u];
                if (idx + 2u < s.nwords) v.z = s.words[idx + 2u];
            }
            *reinterpret_cast<uint4 *>(&win[i]) = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
        }
        if (bmw) {                                       // (g.lo is a multiple of 32: stream words and mark words line up)
            for (uint32_t i = lane; i < WINW; i += 64u) {
                const uint64_t bit = (base + i) * 32u;
                const uint64_t gw = bit >= bm_lo ? (bit - bm_lo) >> 5 : ~0ull;
                bmw[i] = gw < bm_words ? bitmap[gw] : 0u;
            }
        }
        __syncthreads();
        loaded = false;
    }
    // make the register window cover a coded data set at pos (cds() does it itself; marked() wants it done)
    __device__ __forceinline__ void prepare(uint64_t pos)
    {
        if (!loaded || pos < wbase * 32u || pos - wbase * 32u + maxbits + 64u > 2048u) load_regs(pos >> 5);
    }
    // is pos a node of the trunk (after prepare(pos))
    __device__ __forceinline__ bool marked(uint64_t pos) const
    {
        const uint64_t w = pos >> 5;
        return (bmw[w - base] >> (31u - (uint32_t)(pos & 31u))) & 1u;
    }
    __device__ __forceinline__ void load_regs(uint64_t first_word)
    {
        if (base == ~0ull || first_word < base || first_word + 64u > base + WINW) refill(first_word);
        wbase = first_word;
        W = win[first_word - base + lane];
        loaded = true;
    }
    __device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) const
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
    }
    __device__ __forceinline__ uint32_t peek(uint32_t rel) const          // 32 bits at window bit offset rel
    {
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint64_t two = ((uint64_t)rdlane(W, w) << 32) | rdlane(W, (w + 1u) & 63u);
        return (uint32_t)((two << sh) >> 32);
    }
    __device__ __forceinline__ uint32_t scan_incl(uint32_t v) const
    {
        v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);   // row_shr:1
        v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);   // row_shr:2
        v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);   // row_shr:4
        v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);   // row_shr:8
        v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);   // row_bcast:15
        v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);   // row_bcast:31
        return v;
    }
    // window offset just behind the n-th 1 bit at or after rel; 0xFFFFFFFF if the window has fewer
    __device__ __forceinline__ uint32_t skip_ones(uint32_t rel, uint32_t n) const
    {
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint32_t m = lane < w ? 0u : (lane == w ? W & (0xFFFFFFFFu >> sh) : W);
        const uint32_t pc = (uint32_t)__builtin_popcount(m);
        const uint32_t S = scan_incl(pc);
        const uint64_t enough = __ballot(S >= n);
        if (enough == 0) return 0xFFFFFFFFu;
        const uint32_t L = (uint32_t)__builtin_ctzll(enough);
        const uint32_t need = n - (rdlane(S, L) - rdlane(pc, L));          // rank inside word L, 1-based
        const uint32_t word = rdlane(m, L);
        const uint32_t j = lane & 31u;
        const uint32_t bit = (word >> (31u - j)) & 1u;
        const uint32_t rank = j ? (uint32_t)__builtin_popcount(word >> (32u - j)) : 0u;
        const uint64_t hit = __ballot(lane < 32u && bit && rank + 1u == need);
        return L * 32u + (uint32_t)__builtin_ctzll(hit) + 1u;
    }

    // tr_cds(s, c, pos, ref, nz) for a wave-uniform pos: the length in bits of the coded data set that starts at pos
    // (0 = none ends inside the stream), nz = 0 or the zero-run code fs + 1
    __device__ __forceinline__ uint32_t cds(const Cfg &c, uint64_t pos, uint32_t ref, uint32_t &nz)
    {
        nz = 0;
        if (pos + c.id_len >= s.end_bit) return 0;
        prepare(pos);
        const uint32_t rel = (uint32_t)(pos - wbase * 32u);
        const uint32_t h = peek(rel);
        const uint32_t id = h >> (32u - c.id_len);
        uint32_t q = rel + c.id_len;
        if (id == 0u) {
            const uint32_t sel = (h >> (31u - c.id_len)) & 1u;
            q += 1u + ref * c.bps;
            if (sel) {
                q = skip_ones(q, c.bs / 2u);
            } else {
                const uint32_t e = skip_ones(q, 1u);
                if (e != 0xFFFFFFFFu) nz = e - q;
                q = e;
            }
        } else if (id == idmax) {
            q += c.bs * c.bps;
        } else {
            q += ref * c.bps;
            q = skip_ones(q, c.bs - ref);
            if (q != 0xFFFFFFFFu) q += (c.bs - ref) * (id - 1u);
        }
        if (q == 0xFFFFFFFFu) return tr_cds(s, c, pos, ref, nz);          // (beyond the register window: from memory)
        const uint32_t len = q - rel;
        if (pos + len > s.end_bit) {
            nz = 0;
            return 0;
        }
        return len;
    }
};

}  // namespace aec
