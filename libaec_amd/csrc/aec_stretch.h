// aec_stretch.h -- one WAVEFRONT walks a chain of coded data sets 64 BITS AT A TIME (device only; aec_idx.hip).
//
// A chain of coded data sets is serial (where one ends the next begins, reference src/decode.c:402-421), and a lane
// that follows one through device memory pays ~280 dependent vector instructions and a memory round trip per coded
// data set, 2.5 us (aec_trunk.h: tr_cds); a wavefront that parses ONE coded data set cooperatively (aec_coop.h) pays
// its cross-lane round trips one behind the other, ~1 us.  Here the wavefront stages a window of the stream in LDS
// and, for a PIECE of it, the positions of its 1-bits and a rank per word; with those "the coded data set that would
// begin at bit q" is ~50 vector instructions WITHOUT a dependence between lanes -- the end of a unary part is the n-th
// 1-bit behind the header, ones[rank(q1) + n - 1] (aec_spec.h) -- so the 64 lanes parse the coded data sets that would
// begin at 64 CONSECUTIVE bits at once, with and without a reference sample, and the chain through those 64 bits is
// one lane read per coded data set (~0.5 us per 64 bits + 0.05 us per coded data set).  The same tables also serve 64
// chains of a wavefront that stand near each other (k_lock_guess): every lane parses at its OWN position.
// Whatever the tables do not resolve (a coded data set longer than an encoder writes, the end of the stream) is the
// caller's to take with tr_cds: entries are exact or 0, never wrong.
#pragma once

#include "aec_trunk.h"

namespace aec {

constexpr uint32_t kSwWin = 1024;          // words of stream per wavefront in LDS
constexpr uint32_t kSwPad = 8;             // zero words behind them
constexpr uint32_t kSwPiece = 2048;        // bits of a piece (WaveStream; WaveStreamT<P> for other sizes)
constexpr uint32_t kSwLookWords = 68;      // + the longest coded data set of an encoder (2118 bits) and slack
constexpr uint32_t kSwPieceWords = kSwPiece / 32 + kSwLookWords;
// LDS words of one wavefront: window | rank per word of the piece (u16) | positions of the piece's 1-bits (u16)
constexpr uint32_t sw_wave_words(uint32_t piece_bits)
{
    return kSwWin + kSwPad + (piece_bits / 32 + kSwLookWords + 2 + 1) / 2 + (piece_bits / 32 + kSwLookWords) * 16;
}
constexpr uint32_t kSwWaveWords = sw_wave_words(kSwPiece);

template <uint32_t PIECE>
struct WaveStreamT {
    static constexpr uint32_t kPiece = PIECE, kPieceWords = PIECE / 32 + kSwLookWords;
    uint32_t *win;
    uint16_t *prank, *ones;
    const uint32_t *words;
    uint64_t nwords, end_bit;
    uint64_t base;                 // stream word of win[0]; ~0: nothing loaded
    uint32_t pc0, tcnt, plim;      // piece: first bit (window-relative, multiple of 32), its 1-bits, stream bits of the window
    uint32_t wlim;                 // stream bits of the window (set by refill; plim is the piece's copy)
    bool piece_ok;
    uint32_t lane, idmax, maxbits;

    __device__ __forceinline__ void init(uint32_t *lds, const TrStream &s, const Cfg &c)
    {
        win = lds;
        prank = reinterpret_cast<uint16_t *>(lds + kSwWin + kSwPad);
        ones = reinterpret_cast<uint16_t *>(lds + kSwWin + kSwPad + (kPieceWords + 2 + 1) / 2);
        words = s.words;
        nwords = s.nwords;
        end_bit = s.end_bit;
        base = ~0ull;
        pc0 = tcnt = plim = wlim = 0;
        piece_ok = false;
        lane = threadIdx.x & 63u;
        idmax = (1u << c.id_len) - 1u;
        maxbits = c.id_len + 1u + c.bps + c.bs * c.bps;
    }
    // does the scheme serve this parameter set (a coded data set of the encoder's kind inside the look-ahead)?
    __device__ __forceinline__ bool usable() const { return maxbits <= (kSwLookWords - 2u) * 32u; }

    __device__ __forceinline__ void sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
    __device__ __forceinline__ void refill(uint64_t from_word)
    {
        base = from_word & ~3ull;
        sync();
        for (uint32_t i = lane * 4u; i < kSwWin; i += 64u * 4u) {
            const uint64_t at = base + i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (at + 4u <= nwords) {
                v = *reinterpret_cast<const uint4 *>(words + at);
            } else {
                if (at < nwords) v.x = words[at];
                if (at + 1u < nwords) v.y = words[at + 1u];
                if (at + 2u < nwords) v.z = words[at + 2u];
            }
            *reinterpret_cast<uint4 *>(&win[i]) = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
        }
        if (lane < kSwPad) win[kSwWin + lane] = 0u;
        sync();
        piece_ok = false;
        const uint64_t wbits = (uint64_t)kSwWin * 32u;
        const uint64_t sbits = end_bit > base * 32u ? end_bit - base * 32u : 0u;
        wlim = (uint32_t)(sbits < wbits ? sbits : wbits);
    }
    __device__ __forceinline__ void build_piece(uint32_t from_bit)
    {
        pc0 = from_bit & ~31u;
        const uint32_t w0 = pc0 >> 5;
        const uint64_t wbits = (uint64_t)kSwWin * 32u;
        const uint64_t sbits = end_bit > base * 32u ? end_bit - base * 32u : 0u;
        plim = (uint32_t)(sbits < wbits ? sbits : wbits);
        sync();
        uint32_t carry = 0;
        for (uint32_t i0 = 0; i0 < kPieceWords; i0 += 64u) {
            const uint32_t i = i0 + lane, wi = w0 + i;
            const uint32_t word = (i < kPieceWords && wi < kSwWin) ? win[wi] : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(word);
            const uint32_t incl = scan_incl(pc);
            if (i < kPieceWords) prank[i + 1u] = (uint16_t)(carry + incl);
            uint32_t at = carry + incl - pc, bits = word;
            const uint32_t bbase = wi * 32u + 1u;
            while (bits) {
                const uint32_t z = (uint32_t)__builtin_clz(bits);
                bits &= ~(0x80000000u >> z);
                ones[at++] = (uint16_t)(bbase + z);
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) prank[0] = 0;
        tcnt = carry;
        sync();
        piece_ok = true;
    }
    // Make window and piece cover the positions [pos, pos + span) with their look-ahead; returns pos relative to the
    // window.  span <= kPiece / 2.
    __device__ __forceinline__ uint32_t ensure(uint64_t pos, uint32_t span)
    {
        const uint64_t w = pos >> 5;
        bool reload = base == ~0ull || w < base;
        if (!reload && base + kSwWin < nwords) {        // (a window that reaches the end of the stream stays)
            const uint64_t at = pos - base * 32u;
            const bool rebuild = !piece_ok || at < pc0 || at + span > pc0 + kPiece;
            reload = at + span + kSwLookWords * 32u > kSwWin * 32u ||
                     (rebuild && (at & ~31ull) + kPiece + kSwLookWords * 32u > kSwWin * 32u);   // a whole piece, not its head
        }
        if (reload) refill(w);
        const uint32_t rel = (uint32_t)(pos - base * 32u);
        if (!piece_ok || rel < pc0 || rel + span > pc0 + kPiece) build_piece(rel);
        return rel;
    }
    // the option of the coded data set that would begin at window bit q (its first id_len bits)
    __device__ __forceinline__ uint32_t option(const Cfg &c, uint32_t q) const
    {
        const uint32_t w = q >> 5, sh = q & 31u;
        const uint32_t wa = w < kSwWin + kSwPad - 1u ? w : kSwWin + kSwPad - 2u;
        const uint32_t h = (uint32_t)(((((uint64_t)win[wa]) << 32) | win[wa + 1u]) << sh >> 32);
        return h >> (32u - c.id_len);
    }
    // Entry (aec_spec.h nxt[] format: length | kNxtBlock or kNxtZero; 0 = not resolved here) of the coded data set that
    // would begin at window bit q, pc0 <= q < pc0 + kPiece; ref: with a reference sample behind its header.
    __device__ __forceinline__ uint32_t entry(const Cfg &c, uint32_t q, uint32_t ref) const
    {
        const uint32_t il = c.id_len, w = q >> 5, sh = q & 31u;
        const uint32_t wa = w < kSwWin + kSwPad - 1u ? w : kSwWin + kSwPad - 2u;
        const uint32_t a = win[wa], bw = win[wa + 1u];
        const uint32_t h = (uint32_t)(((((uint64_t)a) << 32) | bw) << sh >> 32);
        const uint32_t id = h >> (32u - il);
        const bool unc = id == idmax, low = id == 0u;
        const uint32_t selb = (h >> (31u - il)) & 1u;
        const uint32_t q1 = q + il + (low ? 1u : 0u) + ((ref && !unc) ? c.bps : 0u);
        const uint32_t n = low ? (selb ? c.bs / 2u : 1u) : c.bs - ref;
        const uint32_t w1 = q1 >> 5, sh1 = q1 & 31u;
        const uint32_t w1c = w1 < kSwWin + kSwPad ? w1 : kSwWin + kSwPad - 1u;
        const uint32_t pi = w1 - (pc0 >> 5);
        const uint32_t r1 = (uint32_t)prank[pi < kPieceWords ? pi : kPieceWords] +
                            (sh1 ? (uint32_t)__builtin_popcount(win[w1c] >> (32u - sh1)) : 0u);
        const uint32_t k = r1 + n - 1u;
        const uint32_t e = (k < tcnt && n != 0u) ? (uint32_t)ones[k] : 0u;
        const uint32_t add = low ? 0u : n * (id - 1u);
        const uint32_t end = unc ? q + il + c.bs * c.bps : e + add;
        const bool ok = (unc || e != 0u) && q1 < plim && end <= plim && end - q < 4096u;
        return ok ? ((end - q) | ((low && !selb) ? kNxtZero : kNxtBlock)) : 0u;
    }

    // ---- long coded data sets: HALF A WAVEFRONT parses ONE coded data set out of the window, no piece tables ----
    // The tables above pay where coded data sets are short (64 consecutive bits hold several boundaries, and a piece
    // of 2048 bits is built once for dozens of them); where one is hundreds of bits long the piece is rebuilt every
    // third coded data set (~7 us) to look up three 1-bits.  Here the 32 lanes of a half take the 32 words behind the
    // header: popcount, prefix sum over the half, the word in which the n-th 1-bit lies, the bit in it -- one LDS read
    // and ~60 vector instructions, and the two halves of the wavefront parse two coded data sets at once (with and
    // without a reference sample; two chains).  1024 bits of unary part; beyond: 0, the caller's tr_cds.

    // the window alone covers [pos, pos + span) and the look-ahead behind; returns pos relative to the window
    __device__ __forceinline__ uint32_t ensure_win(uint64_t pos, uint32_t span)
    {
        const uint64_t w = pos >> 5;
        bool reload = base == ~0ull || w < base;
        if (!reload && base + kSwWin < nwords)
            reload = (pos - base * 32u) + span + kSwLookWords * 32u > kSwWin * 32u;
        if (reload) refill(w);
        return (uint32_t)(pos - base * 32u);
    }
    __device__ __forceinline__ uint32_t scan_incl32(uint32_t v) const        // prefix sums over each half of the wavefront
    {
        v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);   // row_shr:1
        v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);   // row_shr:2
        v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);   // row_shr:4
        v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);   // row_shr:8
        v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);   // row_bcast:15
        return v;
    }
    // index (0 = the most significant bit) of the need-th 1-bit of w, 1 <= need <= popcount(w)
    static __device__ __forceinline__ uint32_t sel_msb(uint32_t w, uint32_t need)
    {
        uint32_t pos = 0, t = (uint32_t)__builtin_popcount(w >> 16);
        if (need > t) {
            need -= t;
            pos = 16u;
        }
        uint32_t x = (w >> (16u - pos)) & 0xFFFFu;
        t = (uint32_t)__builtin_popcount(x >> 8);
        if (need > t) {
            need -= t;
            pos += 8u;
            x &= 0xFFu;
        } else {
            x >>= 8;
        }
        t = (uint32_t)__builtin_popcount(x >> 4);
        if (need > t) {
            need -= t;
            pos += 4u;
            x &= 0xFu;
        } else {
            x >>= 4;
        }
        t = (uint32_t)__builtin_popcount(x >> 2);
        if (need > t) {
            need -= t;
            pos += 2u;
            x &= 3u;
        } else {
            x >>= 2;
        }
        if (need > (x >> 1)) pos += 1u;
        return pos;
    }
    // Entry (as entry()) of the coded data set that would begin at window bit q, parsed by the lane's HALF of the
    // wavefront: q and ref are the same in the 32 lanes of a half (and may differ between the halves); every lane of
    // the half gets the result.  q + kSwLookWords * 32 inside the window (ensure_win), or q >= the window's end: 0.
    __device__ __forceinline__ uint32_t coop_half(const Cfg &c, uint32_t q, uint32_t ref) const
    {
        const uint32_t il = c.id_len, w = q >> 5, sh = q & 31u;
        const uint32_t wa = w < kSwWin + kSwPad - 1u ? w : kSwWin + kSwPad - 2u;
        const uint32_t h = (uint32_t)(((((uint64_t)win[wa]) << 32) | win[wa + 1u]) << sh >> 32);
        const uint32_t id = h >> (32u - il);
        const bool unc = id == idmax, low = id == 0u;
        const uint32_t selb = (h >> (31u - il)) & 1u;
        const uint32_t q1 = q + il + (low ? 1u : 0u) + ((ref && !unc) ? c.bps : 0u);
        const uint32_t n = low ? (selb ? c.bs / 2u : 1u) : c.bs - ref;
        const uint32_t l32 = lane & 31u;
        const uint32_t wi = (q1 >> 5) + l32, sh1 = q1 & 31u;
        uint32_t word = wi < kSwWin + kSwPad ? win[wi] : 0u;
        if (l32 == 0u && sh1) word &= 0xFFFFFFFFu >> sh1;
        const uint32_t pc = (uint32_t)__builtin_popcount(word);
        const uint32_t incl = scan_incl32(pc);
        const bool hit = incl >= n && incl - pc < n && n != 0u;
        const uint64_t m = __ballot(hit);
        const uint32_t mh = (uint32_t)(m >> (lane & 32u));
        const uint32_t at = hit ? wi * 32u + sel_msb(word, n - (incl - pc)) + 1u : 0u;
        const uint32_t src = (lane & 32u) + (mh ? (uint32_t)__builtin_ctz(mh) : 0u);
        const uint32_t e = mh ? (uint32_t)__shfl((int)at, (int)src) : 0u;
        const uint32_t add = low ? 0u : n * (id - 1u);
        const uint32_t end = unc ? q + il + c.bs * c.bps : e + add;
        const bool ok = (unc || e != 0u) && q < wlim && q1 < wlim && end <= wlim && end - q < 4096u;
        return ok ? ((end - q) | ((low && !selb) ? kNxtZero : kNxtBlock)) : 0u;
    }
};

typedef WaveStreamT<kSwPiece> WaveStream;

}  // namespace aec
