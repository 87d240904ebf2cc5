// aec_idx.hip -- RSI index pass for streams that arrive without an offset table (gfx950).
//
// A coded data set can only be located by parsing its predecessor (the stream has no lengths,
// reference src/decode.c:402-421), so finding where the RSIs start is a serial walk in the
// reference.  Here it is split into
//   k_spec    SPECULATION, fully parallel: a workgroup takes a window of the stream into LDS,
//             builds rank/select over its 1-bits, tabulates "a CDS starts at bit q: how long is it"
//             for every q of the window (aec_spec.h), then tries every bit position of the window's
//             core as an RSI start (ref CDS + table hops over rsi blocks) and finally chains those
//             RSI hops until they leave the core.  Output per bit position p: T[p] = length of an
//             RSI starting at p, Xb/Xc[p] = bits / RSIs of the chained hop out of the window.
//             Nearly all of that work is thrown away -- only the entries at true RSI starts are
//             ever read -- but it is what turns the walk into table lookups.
//   k_index   the WALK, one wavefront: at an RSI start it hops over the tables (one lookup per
//             window); an RSI the tables did not resolve (longer than the look-ahead, cut by the
//             end of the stream, malformed) is walked CDS by CDS as before: the stream is served
//             from an LDS window, unary parts are skipped cooperatively (popcount per lane + DPP scan).
//   k_expand  fills in the RSI starts inside the chained hops, one lane per hop.
// Long RSIs (raw RSI bits beyond the LDS look-ahead) take the serial walk alone.  The batch form
// runs one serial walker per independent chunk.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "aec_kernels.h"
#include "aec_lane.h"
#include "aec_spec.h"
#include "aec_spec2.h"

namespace aec {

namespace {

// wave-wide inclusive prefix sum on the DPP network (same sequence as aec_enc.hip wave_scan_dpp)
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}

// Speculative tables of one chunk of the stream: entries for bit positions [lo, hi).
struct IdxTables {
    const uint16_t *T;     // RSI length if an RSI starts here (0 = unresolved)
    const uint16_t *Xb;    // chained hop out of the window core: bits ...
    const uint8_t *Xc;     // ... and RSIs covered (0 = none)
    uint64_t lo, hi;
};

// Sparse tables of the second-generation speculation (k_spec2): candidates = coded-data-set boundaries
// found by self-synchronising chains.  Per core window k (core bits each, the first one at bit lo):
// a bitmap over its bit positions, per bitmap word the number of candidates in front of it (inside the
// window), and up to cap records {RSI length if an RSI starts here, chained hop out of the window}.
struct SparseTables {
    const uint32_t *bitmap;    // bit (31 - i % 32) of word i / 32 <=> position lo + i is a candidate
    const uint16_t *pre;       // per bitmap word: candidates of ITS window in front of the word
    const uint2 *rec;          // [window * cap + index]: x = RSI length (0 = unresolved), y = chain (cnt << 24 | bits)
    const uint16_t *cpos;      // [window * cap + index]: position inside the window core
    const uint32_t *ccnt;      // [window]: candidates
    uint64_t lo, hi;           // tabulated bit range
    uint32_t core, cap;        // bits per window (multiple of 32), record capacity per window
    // wide walker: per chunk of wpc windows, what every candidate of the chunk's first window leads to
    const uint4 *wide;         // [chunk * cap + index]: {exit lo, exit hi, RSIs, 1 = resolved}
    uint32_t wpc;
};

// record of the candidate at absolute bit p (false: p is not a candidate)
__device__ __forceinline__ bool sparse_lookup(const SparseTables &t, uint64_t p, uint2 &rec, uint32_t &window,
                                              uint32_t &index)
{
    const uint64_t i = p - t.lo;
    const uint32_t word = t.bitmap[i >> 5], sh = (uint32_t)(i & 31u);
    if (!((word >> (31u - sh)) & 1u)) return false;
    window = (uint32_t)(i / t.core);
    index = (uint32_t)t.pre[i >> 5] + (sh ? (uint32_t)__popc(word >> (32u - sh)) : 0u);
    rec = t.rec[(uint64_t)window * t.cap + index];
    return true;
}

// Dense hop tables for streams whose RSIs are far longer than any window (k_hops / k_hop_compose):
// "from the coded data set that starts at bit p, where are you 16 / 64 / 256 coded data sets on, and how
// many blocks beyond that count did zero-block runs cover?"  Independent of the RSI structure -- the
// walker itself knows its position inside the RSI, parses the coded data set that carries the reference
// sample and every rest-of-segment run on its own (hops never contain one), and takes the widest hop that
// stays inside the RSI.  h16: bits [0,13) distance, [13,16) extra blocks; h64 / h256: [0,24) distance,
// [24,32) extra blocks; 0 = no entry.
// The base level is 16 coded data sets, or 4 where 16 of them do not fit the 13-bit distance of an h16 entry
// (blocks of 64 16-bit samples: 700 bits per coded data set); n0 says which, the composed levels are 4 n0
// and 16 n0.
struct HopTables {
    const uint16_t *h16;
    const uint32_t *h64, *h256;
    uint64_t lo, hi;
    uint32_t n0;
};

struct ChunkEntry {        // where the true chain enters a chunk the walker skipped over the wide table
    uint64_t pos, r;
    uint32_t valid, pad;
};

struct IdxHop {            // a chained hop the walker took: k_expand writes its RSI starts
    uint64_t pos, r;
    uint32_t cnt, pad;
};

struct IdxCarry {          // walker state between the table chunks of one stream
    uint64_t good, r;
    uint32_t active, n_hops;
    uint64_t cur_start;    // hop tables: a chunk may end inside an RSI -- where that RSI began ...
    uint32_t b, pad;       // ... and the blocks of it in front of `good`
};

// ---- speculation ------------------------------------------------------------------------------------
// LDS: win[nw + 2] u32 | rank[nw + 2] u16 | sel[nw + 2] u16 | nxt[W] | hop4[W] | hop16[W] | Tl[core] (u16)
__global__ void __launch_bounds__(1024)
k_spec(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, uint64_t tab_lo,
       uint32_t core, uint32_t look, uint16_t *__restrict__ T, uint16_t *__restrict__ Xb,
       uint8_t *__restrict__ Xc)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t spec_lds[];
    const uint32_t W = core + look, nw = W / 32u;
    uint32_t *win = spec_lds;
    uint16_t *rank = reinterpret_cast<uint16_t *>(win + nw + 2);
    uint16_t *sel = rank + nw + 2;
    uint16_t *nxt = sel + nw + 2;
    uint16_t *hop4 = nxt + W;
    uint16_t *hop16 = hop4 + W;
    uint16_t *Tl = hop16 + W;
    const uint64_t rel0 = (uint64_t)blockIdx.x * core;
    const uint64_t wstart = tab_lo + rel0;
    if (wstart >= end_bit) return;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint64_t w0 = wstart >> 5;
    for (uint32_t i = tid; i < nw + 2; i += nt) {
        const uint64_t idx = w0 + i;
        win[i] = idx < nwords ? bswap32(words[idx]) : 0u;
    }
    __syncthreads();
    if (tid < 64) {                                   // prefix count of 1-bits per word
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nw; base += 64) {
            const uint32_t i = base + tid;
            const uint32_t pc = i < nw ? (uint32_t)__popc(win[i]) : 0u;
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < nw) rank[i + 1] = (uint16_t)(carry + incl);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (tid == 0) rank[0] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += nt) {         // sampled select: word of every 32nd 1-bit
        const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
        if (32u * m + 1u > lo && 32u * m + 1u <= hi) sel[m] = (uint16_t)i;
    }
    __syncthreads();
    const uint64_t left = end_bit - wstart;
    const SpecWin s{win, rank, sel, nw, left < W ? (uint32_t)left : W};
    // several positions per lane and iteration: the chains of dependent LDS lookups are independent
    // between positions and the straight-line code lets the scheduler overlap them
    for (uint32_t q = tid; q < W; q += 2 * nt) {
        const uint32_t qb = q + nt;
        const uint16_t ea = q < s.limit ? spec_nxt_entry(s, c, q) : (uint16_t)0;
        const uint16_t eb = qb < s.limit ? spec_nxt_entry(s, c, qb) : (uint16_t)0;
        nxt[q] = ea;
        if (qb < W) nxt[qb] = eb;
    }
    __syncthreads();
    for (uint32_t q = tid; q < W; q += 4 * nt) {
        uint16_t e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = q + u * nt < W ? spec_hop4(nxt, c, s.limit, q + u * nt) : (uint16_t)0;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (q + u * nt < W) hop4[q + u * nt] = e[u];
    }
    // first CDS of a hypothetical RSI at every core position (carries the reference sample);
    // parked in Tl until the walk of that position replaces it by the RSI length
    for (uint32_t q = tid; q < core; q += nt) Tl[q] = q < s.limit ? spec_first_entry(s, c, q) : (uint16_t)0;
    __syncthreads();
    for (uint32_t q = tid; q < W; q += 4 * nt) {
        uint16_t e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = q + u * nt < W ? spec_hop16(hop4, s.limit, q + u * nt) : (uint16_t)0;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (q + u * nt < W) hop16[q + u * nt] = e[u];
    }
    __syncthreads();
    // RSI walks.  Walk lengths differ wildly between lanes (hypotheses die early or run the full
    // rsi blocks), so the position loop and the step loop are flattened: every lane keeps kSlots
    // walks in flight (independent LDS lookups per iteration) and a slot that finishes takes the
    // lane's next position at once -- a wave is as slow as its lanes' SUMS of steps, not the sum of
    // per-position maxima.
    {
        constexpr int kSlots = 4;
        const uint32_t per = core / nt;                 // positions per lane: q = tid + k * nt
        uint32_t wp[kSlots], wpos[kSlots], wb[kSlots], wk[kSlots];
        bool act[kSlots];
        auto start = [&](int j) {                       // next position of slot j (k = j, j + kSlots, ...)
            act[j] = false;
            while (wk[j] < per) {
                const uint32_t q = tid + wk[j] * nt;
                wk[j] += kSlots;
                if (spec_walk_init(c, Tl[q], q, wpos[j], wb[j])) {
                    wp[j] = q;
                    act[j] = true;
                    return;
                }
                Tl[q] = 0;
            }
        };
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            wk[j] = (uint32_t)j;
            start(j);
        }
        for (;;) {
            bool any = false;
#pragma unroll
            for (int j = 0; j < kSlots; j++) {
                if (!act[j]) continue;
                any = true;
                bool done = wb[j] >= c.rsi, ok = true;
                if (!done) {
                    ok = spec_walk_step(c, nxt, hop4, hop16, s.limit, wpos[j], wb[j]);
                    done = !ok || wb[j] >= c.rsi;
                }
                if (done) {
                    uint32_t t = ok ? wpos[j] - wp[j] : 0u;
                    if (t > 0xFFFFu) t = 0;
                    Tl[wp[j]] = (uint16_t)t;      // (global T is written from Tl below, coalesced)
                    start(j);
                }
            }
            if (!any) break;
        }
    }
    __syncthreads();
    const bool pad = c.flags & F_PAD_RSI;
    for (uint32_t q = tid; q < core; q += nt) {
        uint32_t pos = q, cnt = 0;
        while (pos < core && pos < s.limit && Tl[pos] && cnt < 255u) {
            pos += Tl[pos];
            cnt++;
            if (pad) pos = (pos + 7u) & ~7u;
        }
        T[rel0 + q] = Tl[q];
        Xb[rel0 + q] = (uint16_t)(pos - q);
        Xc[rel0 + q] = (uint8_t)cnt;
    }
}

// ---- hop tables (long RSIs) ---------------------------------------------------------------------------
// LDS: win[nw + 2] u32 | rank[nw + 2] u16 | sel[nw + 2] u16 | nxt[W] u16 | hop4[W] u16
__global__ void __launch_bounds__(1024)
k_hops(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, uint64_t tab_lo,
       uint32_t core, uint32_t look, uint16_t *__restrict__ h16, uint32_t n0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t spec_lds[];
    const uint32_t W = core + look, nw = W / 32u;
    uint32_t *win = spec_lds;
    uint16_t *rank = reinterpret_cast<uint16_t *>(win + nw + 2);
    uint16_t *sel = rank + nw + 2;
    uint16_t *nxt = sel + nw + 2;
    uint16_t *hop4 = nxt + W;
    const uint64_t rel0 = (uint64_t)blockIdx.x * core;
    const uint64_t wstart = tab_lo + rel0;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    if (wstart >= end_bit) {
        for (uint32_t q = tid; q < core; q += nt) h16[rel0 + q] = 0;
        return;
    }
    const uint64_t w0 = wstart >> 5;
    for (uint32_t i = tid; i < nw + 2; i += nt) {
        const uint64_t idx = w0 + i;
        win[i] = idx < nwords ? bswap32(words[idx]) : 0u;
    }
    __syncthreads();
    if (tid < 64) {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nw; base += 64) {
            const uint32_t i = base + tid;
            const uint32_t pc = i < nw ? (uint32_t)__popc(win[i]) : 0u;
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < nw) rank[i + 1] = (uint16_t)(carry + incl);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (tid == 0) rank[0] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += nt) {
        const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
        if (32u * m + 1u > lo && 32u * m + 1u <= hi) sel[m] = (uint16_t)i;
    }
    __syncthreads();
    const uint64_t left = end_bit - wstart;
    const SpecWin s{win, rank, sel, nw, left < W ? (uint32_t)left : W};
    for (uint32_t q = tid; q < W; q += nt) nxt[q] = q < s.limit ? spec_nxt_entry(s, c, q) : (uint16_t)0;
    __syncthreads();
    for (uint32_t q = tid; q < W; q += nt) hop4[q] = spec_hop4(nxt, c, s.limit, q);
    __syncthreads();
    for (uint32_t q = tid; q < core; q += nt) h16[rel0 + q] = n0 == 16u ? spec_hop16(hop4, s.limit, q) : hop4[q];
}

// four hops of `src` in a row: dst[p] = where they lead (SRC16: src holds h16 entries, else h64 entries)
template <bool SRC16>
__global__ void __launch_bounds__(256)
k_hop_compose(const void *__restrict__ src, uint32_t *__restrict__ dst, uint64_t n)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint64_t pos = p;
    uint32_t extra = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ok = ok && pos < n;
        uint32_t bits, ex;
        if (SRC16) {
            const uint32_t e = static_cast<const uint16_t *>(src)[ok ? pos : p];
            bits = e & kHopBitsMask;
            ex = e >> 13;
            ok = ok && e != 0;
        } else {
            const uint32_t e = static_cast<const uint32_t *>(src)[ok ? pos : p];
            bits = e & 0xFFFFFFu;
            ex = e >> 24;
            ok = ok && e != 0;
        }
        pos += bits;
        extra += ex;
    }
    const uint64_t d = pos - p;
    dst[p] = (ok && d < (1u << 24) && extra < 256u) ? (uint32_t)d | (extra << 24) : 0u;
}

// ---- sparse speculation (aec_spec2.h) ----------------------------------------------------------------
// One workgroup of 1024 lanes per window = [lead-in | core | look-ahead].  LDS:
//   win[nw + 2] u32 | marks[nw] u32 | rank[nw + 2] u16 | sel[nw + 2] u16 | mpre[nw + 2] u16 |
//   cpos[cap] u16 | cnxt[cap] u16 | chop4[cap] u16 | chop16[cap] u16 | ua[cap] u16
// Output for the candidates inside the core: the window's part of the global bitmap / prefix table
// and its records (SparseTables above).
struct Spec2Geom {
    uint32_t lead, core, look, stride, burn, cap_lds, cap_core, budget;
};

__global__ void __launch_bounds__(1024)
k_spec2(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, uint64_t tab_lo,
        uint64_t start_bit, const Spec2Geom g, uint32_t *__restrict__ gbitmap, uint16_t *__restrict__ gpre,
        uint2 *__restrict__ grec, uint16_t *__restrict__ gcpos, uint32_t *__restrict__ gccnt,
        unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t spec_lds[];
    __shared__ uint32_t sh_total;
    // (diagnostic builds of the host pass `prof`: shader-clock stamps at the phase boundaries)
    auto stamp = [&](int k) {
        if (prof && threadIdx.x == 0) prof[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const uint32_t W = g.lead + g.core + g.look, nw = W / 32u, cap = g.cap_lds;
    uint32_t *win = spec_lds;
    uint32_t *marks = win + nw + 2;
    uint16_t *rank = reinterpret_cast<uint16_t *>(marks + nw);
    uint16_t *sel = rank + nw + 2;
    uint16_t *mpre = sel + nw + 2;
    uint16_t *cpos = mpre + nw + 2;
    uint16_t *cnxt = cpos + cap;
    uint16_t *chop4 = cnxt + cap;
    uint16_t *chop16 = chop4 + cap;
    uint16_t *ua = chop16 + cap;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint64_t core_abs = tab_lo + (uint64_t)blockIdx.x * g.core;
    const uint64_t gw0 = ((uint64_t)blockIdx.x * g.core) >> 5;              // first bitmap word of this window
    const uint32_t cw = g.core / 32u;
    auto give_up = [&]() {                                                  // nothing tabulated: the serial walk
        for (uint32_t i = tid; i < cw; i += nt) {
            gbitmap[gw0 + i] = 0u;
            gpre[gw0 + i] = 0;
        }
        if (tid == 0) gccnt[blockIdx.x] = 0u;
    };
    if (core_abs >= end_bit) {
        give_up();
        return;
    }
    const uint64_t wstart = core_abs >= g.lead ? core_abs - g.lead : 0;     // multiple of 32
    const uint32_t c0 = (uint32_t)(core_abs - wstart), c1 = c0 + g.core;
    const uint64_t w0 = wstart >> 5;
    for (uint32_t i = tid; i < nw + 2; i += nt) {
        const uint64_t idx = w0 + i;
        win[i] = idx < nwords ? bswap32(words[idx]) : 0u;
        if (i < nw) marks[i] = 0u;
    }
    __syncthreads();
    // prefix counts of 1-bits (and, further down, of marks) per word: one wave, 64 words per round
    auto prefix16 = [&](const uint32_t *src, uint16_t *dst) {
        if (tid < 64) {
            uint32_t carry = 0;
            for (uint32_t base = 0; base < nw; base += 64) {
                const uint32_t i = base + tid;
                const uint32_t pc = i < nw ? (uint32_t)__popc(src[i]) : 0u;
                const uint32_t incl = wave_incl_sum_dpp(pc);
                if (i < nw) dst[i + 1] = (uint16_t)(carry + incl);
                carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (tid == 0) {
                dst[0] = 0;
                sh_total = carry;
            }
        }
    };
    prefix16(win, rank);
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += nt) {         // sampled select: word of every 32nd 1-bit
        const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
        if (32u * m + 1u > lo && 32u * m + 1u <= hi) sel[m] = (uint16_t)i;
    }
    __syncthreads();
    const uint64_t left = end_bit - wstart;
    const SpecWin s{win, rank, sel, nw, left < W ? (uint32_t)left : W};
    stamp(1);

    // ---- 1. sync chains: burn in, then mark until a marked boundary is met
    if (tid == 0 && start_bit >= wstart && start_bit - wstart < s.limit) {
        const uint32_t q = (uint32_t)(start_bit - wstart);
        atomicOr(&marks[q >> 5], 1u << (31u - (q & 31u)));                  // the one boundary that is known
    }
    for (uint32_t q0 = tid * g.stride; q0 < s.limit; q0 += nt * g.stride) {
        uint32_t q = q0;
        bool ok = true;
        for (uint32_t k = 0; k < g.burn && ok; k++) {
            const uint32_t len = s2_chain_step(s, c, q);
            ok = len != 0;
            q += len;
        }
        while (ok && q < s.limit) {
            const uint32_t bit = 1u << (31u - (q & 31u));
            if (atomicOr(&marks[q >> 5], bit) & bit) break;
            const uint32_t len = s2_chain_step(s, c, q);
            ok = len != 0;
            q += len;
        }
    }
    __syncthreads();
    stamp(2);
    prefix16(marks, mpre);
    __syncthreads();
    const uint32_t ncand = sh_total;
    if (ncand > cap) {                       // (a stream of minimal coded data sets: one candidate per few bits)
        give_up();
        return;
    }
    // ---- 2. tables on the candidates (positions first, so that the CDS parse below runs one
    // candidate per lane instead of one bitmap word per lane)
    for (uint32_t i = tid; i < nw; i += nt) {
        uint32_t m = marks[i], idx = mpre[i];
        while (m) {
            const uint32_t b = (uint32_t)__builtin_clz(m);
            m &= ~(0x80000000u >> b);
            cpos[idx++] = (uint16_t)(i * 32u + b);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) {
        const uint32_t q = cpos[i];
        cnxt[i] = q < s.limit ? spec_nxt_entry(s, c, q) : (uint16_t)0;
    }
    __syncthreads();
    stamp(3);
    S2Win w{s, marks, mpre, cnxt, chop4, chop16, ncand};
    for (uint32_t i = tid; i < ncand; i += nt) chop4[i] = s2_hop4(w, c, cpos, i);
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) chop16[i] = s2_hop16(w, cpos, i);
    __syncthreads();
    stamp(4);
    // ---- 3. the RSI hypothesis at every candidate of the core
    const uint32_t i0 = mpre[c0 >> 5], i1 = mpre[c1 >> 5 < nw ? c1 >> 5 : nw];   // (c0, c1 are multiples of 32)
    const uint32_t ncore = i1 - i0;
    // (Flattening this loop -- one step per lane and iteration, a finished lane taking its next candidate
    // at once -- was measured and is SLOWER here, 521k against 430k cycles per window: the iterations of
    // the merged loop pay for every path, the on-demand parse included.)
    for (uint32_t i = i0 + tid; i < i1; i += nt) {
        const uint32_t q = cpos[i];
        uint32_t a = q < s.limit ? s2_unit(w, c, q, 0u, c.rsi, g.budget) : 0u;
        if (a > 0xFFFFu) a = 0;
        ua[i] = (uint16_t)a;
    }
    __syncthreads();
    stamp(5);
    if (ncore > g.cap_core) {
        give_up();
        return;
    }
    // ---- 4. chains out of the core, records, this window's part of the bitmap
    for (uint32_t i = i0 + tid; i < i1; i += nt) {
        uint32_t pos = cpos[i], cnt = 0;
        while (pos < c1 && pos < s.limit && cnt < 255u) {
            const uint32_t j = s2_index(w, pos);
            if (j == kS2NoIndex || !ua[j]) break;
            pos += ua[j];
            cnt++;
        }
        const uint64_t at = (uint64_t)blockIdx.x * g.cap_core + (i - i0);
        grec[at] = make_uint2(ua[i], cnt ? ((cnt << 24) | (pos - cpos[i])) : 0u);
        gcpos[at] = (uint16_t)(cpos[i] - c0);
    }
    for (uint32_t i = tid; i < cw; i += nt) {
        gbitmap[gw0 + i] = marks[(c0 >> 5) + i];
        gpre[gw0 + i] = (uint16_t)(mpre[(c0 >> 5) + i] - i0);
    }
    if (tid == 0) gccnt[blockIdx.x] = ncore;
    stamp(6);
}

// ---- wide walker: every candidate of a chunk's first window chases the window chain through the chunk
__global__ void __launch_bounds__(256)
k_wide(const SparseTables t, uint32_t nwin, uint64_t end_bit, uint4 *__restrict__ wide)
{
    const uint32_t chunk = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t first = chunk * t.wpc;
    if (first >= nwin || i >= t.ccnt[first]) return;
    const uint32_t last = first + t.wpc < nwin ? first + t.wpc : nwin;      // one past the chunk's windows
    const uint64_t stop = t.lo + (uint64_t)last * t.core;
    uint64_t pos = t.lo + (uint64_t)first * t.core + t.cpos[(uint64_t)first * t.cap + i];
    uint32_t cnt = 0, ok = 1;
    while (pos < stop && pos < end_bit) {
        uint2 rec;
        uint32_t wv, ix;
        if (!sparse_lookup(t, pos, rec, wv, ix) || !(rec.y >> 24)) {
            ok = 0;
            break;
        }
        pos += rec.y & 0xFFFFFFu;
        cnt += rec.y >> 24;
    }
    // (a chain that ends at the end of the input inside the last chunk is resolved as far as it goes:
    // the walker takes over from its exit)
    wide[(uint64_t)chunk * t.cap + i] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), cnt, ok && cnt ? 1u : 0u);
}

// the true chain through the chunks the walker skipped: one lane per chunk records the window hops
__global__ void __launch_bounds__(64)
k_rewalk(const SparseTables t, uint32_t nwin, uint32_t nchunks, uint64_t end_bit,
         const ChunkEntry *__restrict__ entry, IdxHop *__restrict__ hops, uint32_t *__restrict__ nhops)
{
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= nchunks) return;
    uint32_t n = 0;
    if (entry[chunk].valid) {
        const uint32_t first = chunk * t.wpc;
        const uint32_t last = first + t.wpc < nwin ? first + t.wpc : nwin;
        const uint64_t stop = t.lo + (uint64_t)last * t.core;
        uint64_t pos = entry[chunk].pos, r = entry[chunk].r;
        IdxHop *out = hops + (uint64_t)chunk * t.wpc * 2u;
        while (pos < stop && pos < end_bit && n < t.wpc * 2u) {
            uint2 rec;
            uint32_t wv, ix;
            if (!sparse_lookup(t, pos, rec, wv, ix) || !(rec.y >> 24)) break;   // (cannot happen: k_wide went through)
            out[n++] = IdxHop{pos, r, rec.y >> 24, 0u};
            pos += rec.y & 0xFFFFFFu;
            r += rec.y >> 24;
        }
    }
    nhops[chunk] = n;
}

// RSI starts inside hops, from the sparse records.  lists == 0: the walker's own hop list (count in
// carry->n_hops); lists > 0: the per-chunk lists of k_rewalk (`stride` entries apart, counts in nhops).
__global__ void k_expand2(const SparseTables t, const IdxCarry *__restrict__ carry, const IdxHop *__restrict__ hops,
                          const uint32_t *__restrict__ nhops, uint32_t lists, uint32_t stride,
                          uint64_t *__restrict__ rsi_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    IdxHop h;
    if (lists == 0) {
        if (i >= carry->n_hops) return;
        h = hops[i];
    } else {
        const uint32_t list = i / stride, j = i % stride;
        if (list >= lists || j >= nhops[list]) return;
        h = hops[(uint64_t)list * stride + j];
    }
    uint64_t p = h.pos;
    for (uint32_t j = 0; j < h.cnt; j++) {
        rsi_off[h.r + j] = p;
        uint2 rec;
        uint32_t wv, ix;
        if (!sparse_lookup(t, p, rec, wv, ix)) return;      // (cannot happen inside a chained hop)
        p += rec.x;
    }
}

// RSI starts inside the chained hops of the walker
__global__ void k_expand(const IdxCarry *__restrict__ carry, const IdxHop *__restrict__ hops,
                         const IdxTables tabs, uint32_t pad_rsi, uint64_t *__restrict__ rsi_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= carry->n_hops) return;
    const IdxHop h = hops[i];
    uint64_t p = h.pos;
    for (uint32_t j = 0; j < h.cnt; j++) {
        rsi_off[h.r + j] = p;
        p += tabs.T[p - tabs.lo];
        if (pad_rsi) p = (p + 7u) & ~7ull;
    }
}

// ---- serial RSI index -----------------------------------------------------------------------------
// One wavefront per stream.  The walk itself is serial (every lane executes it redundantly on
// wave-uniform values), but the stream is served from a 16 KiB LDS window that all 64 lanes refill
// with coalesced 16-byte loads, so the parser never waits for HBM: a dependent global load per
// refill of the bit window held the first version at ~5 MB/s of compressed input.
constexpr uint32_t kIdxWindowWords = 4096;

struct LdsWindowFetch {
    const uint32_t *lds;    // window of kIdxWindowWords words (host order)
    uint64_t base;          // stream word index of lds[0]
    __device__ __forceinline__ uint32_t operator()(uint64_t idx) const
    {
        const uint64_t rel = idx - base;
        return rel < kIdxWindowWords ? lds[rel] : 0u;   // outside: refilled before the next CDS
    }
};

// chunk_off == nullptr: one stream [start_bit, end_bit) -> rsi_off[0..max_rsi), res[0].
// chunk_off != nullptr: workgroup s walks the independent stream that occupies bytes
// [chunk_off[s], chunk_off[s+1]) of the buffer and writes rsi_off[s*max_rsi ..], res[s]; offsets are
// absolute bit positions in the buffer, so ONE k_decode launch decodes the RSIs of all streams.
__global__ void __launch_bounds__(64)
k_index(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
        uint64_t start_bit, uint64_t *__restrict__ rsi_off, uint64_t max_rsi, DecResult *res,
        const uint64_t *__restrict__ chunk_off, const IdxTables tabs, IdxHop *__restrict__ hops,
        uint32_t hop_cap, IdxCarry *carry, uint32_t first, uint32_t last, uint32_t start_block,
        uint64_t rsi_start, uint32_t tail_slot, const SparseTables sp, ChunkEntry *__restrict__ centry,
        const HopTables ht)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[kIdxWindowWords];
    uint64_t r = 0;
    if (chunk_off) {
        start_bit = chunk_off[blockIdx.x] * 8u;
        end_bit = chunk_off[blockIdx.x + 1] * 8u;
        rsi_off += (uint64_t)blockIdx.x * max_rsi;
        res += blockIdx.x;
    } else if (blockIdx.x != 0) {
        return;
    }
    // the first walk of a call starts the result record (there is no separate init launch)
    if (!chunk_off && first && threadIdx.x == 0) {
        res->n_rsi = 0;
        res->tail_blocks = 0;
        res->end_bit = start_bit;
        res->status = DEC_OK;
        res->pad = 0;
        res->bad_rsi = ~0ull;
    }
    if (carry && !first) {               // continue where the walk over the previous table chunk stopped
        if (!carry->active) {
            if (threadIdx.x == 0) carry->n_hops = 0;
            return;
        }
        start_bit = carry->good;
        r = carry->r;
    }
    const uint32_t b_carried = (carry && !first) ? carry->b : 0u;
    const uint64_t start_carried = (carry && !first) ? carry->cur_start : 0ull;
    const uint32_t lane = threadIdx.x;
    const bool pp = c.flags & F_PREPROCESS;
    const uint32_t maxw = (c.id_len + 1 + c.bps + c.bs * c.bps) / 32 + 4;   // words one CDS can touch

    uint64_t base = (start_bit >> 5) & ~3ull;
    auto refill = [&](uint64_t from_word) {
        base = from_word & ~3ull;
        for (uint32_t i = lane * 4; i < kIdxWindowWords; i += 64 * 4) {
            const uint64_t idx = base + i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx + 4 <= nwords) {
                v = *reinterpret_cast<const uint4 *>(words + idx);
            } else {
                if (idx < nwords) v.x = words[idx];
                if (idx + 1 < nwords) v.y = words[idx + 1];
                if (idx + 2 < nwords) v.z = words[idx + 2];
            }
            *reinterpret_cast<uint4 *>(&win[i]) = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
        }
        __syncthreads();
    };
    refill(base);

    // Cooperative walk: the 64 lanes hold 64 consecutive stream words (a 2048-bit window) in
    // registers; locating the end of a CDS is a masked popcount per lane, one DPP prefix sum, a
    // ballot and a rank-select inside one word -- about 60 wave instructions per CDS instead of a
    // bit-serial loop.  Everything below is wave-uniform except W.
    const uint32_t maxbits = c.id_len + 1 + c.bps + c.bs * c.bps;
    const bool coop = maxbits + 128u <= 2048u;
    const uint32_t idmax = (1u << c.id_len) - 1u;
    uint64_t wbase = 0;          // stream word held by lane 0
    uint32_t W = 0;
    auto load_regs = [&](uint64_t first_word) {
        wbase = first_word;
        W = LdsWindowFetch{win, base}(first_word + lane);
    };
    auto rdlane = [&](uint32_t v, uint32_t l) {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
    };
    auto peek = [&](uint32_t rel) {                     // 32 bits at window bit offset rel
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint64_t two = ((uint64_t)rdlane(W, w) << 32) | rdlane(W, (w + 1) & 63u);
        return (uint32_t)((two << sh) >> 32);
    };
    // window offset just behind the n-th 1 bit at or after rel; 0xFFFFFFFF if the window has fewer
    auto skip_ones = [&](uint32_t rel, uint32_t n) -> uint32_t {
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint32_t m = lane < w ? 0u : (lane == w ? W & (0xFFFFFFFFu >> sh) : W);
        const uint32_t pc = (uint32_t)__builtin_popcount(m);
        const uint32_t S = wave_incl_sum_dpp(pc);
        const uint64_t enough = __ballot(S >= n);
        if (enough == 0) return 0xFFFFFFFFu;
        const uint32_t L = (uint32_t)__builtin_ctzll(enough);
        const uint32_t need = n - (rdlane(S, L) - rdlane(pc, L));      // rank inside word L, 1-based
        const uint32_t word = rdlane(m, L);
        const uint32_t j = lane & 31u;
        const uint32_t bit = (word >> (31u - j)) & 1u;
        const uint32_t rank = j ? (uint32_t)__builtin_popcount(word >> (32u - j)) : 0u;
        const uint64_t hit = __ballot(lane < 32u && bit && rank + 1u == need);
        return L * 32u + (uint32_t)__builtin_ctzll(hit) + 1u;
    };

    BitReaderT<LdsWindowFetch> br;
    br.init(LdsWindowFetch{win, base}, end_bit, start_bit);
    uint64_t good = start_bit;
    uint32_t b = 0, status = DEC_OK, nh = 0;
    // Resumed walk (streaming callers): start_bit is a CDS boundary inside an RSI that began at
    // rsi_start and of which start_block blocks lie before start_bit.
    uint64_t cur_start = start_bit;          // start of the RSI being walked
    if (first && !chunk_off && start_block) {
        b = start_block;
        cur_start = rsi_start;
        if (lane == 0 && max_rsi) rsi_off[0] = rsi_start;
    }
    if (b_carried) {             // the previous table chunk ended inside this RSI
        b = b_carried;
        cur_start = start_carried;
    }
    if (coop) load_regs(good >> 5);
    bool stale = false;          // the sequential reader lags behind `good` (hops moved it)
    for (;;) {
        bool hopped = false;
        if (b == 0) {
            if (r >= max_rsi) break;
            if ((c.flags & F_PAD_RSI) && (good & 7u)) {      // reference decode.c:407-408
                good = (good + 7u) & ~7ull;
                hopped = true;
            }
            // Fast hops over the speculative tables (k_spec): at an RSI start one lookup gives the
            // end of a chain of whole RSIs leaving the window (Xb/Xc; the RSI starts inside the hop
            // are filled in by k_expand), or of this RSI alone (T).  An entry of 0 = not resolved
            // by the tables: that RSI is walked CDS by CDS below.
            // Sparse tables (k_spec2): at a chunk's first window ONE lookup in the wide walker's table
            // takes the walk across the whole chunk (k_rewalk / k_expand2 fill in what lies inside);
            // else the window's chained hop, else this RSI alone.
            while (sp.bitmap && r < max_rsi && good >= sp.lo && good < sp.hi && good < end_bit) {
                uint2 rec;
                uint32_t wv, ix;
                if (!sparse_lookup(sp, good, rec, wv, ix)) break;
                if (sp.wide && (wv % sp.wpc) == 0u) {
                    const uint4 wd = sp.wide[(uint64_t)(wv / sp.wpc) * sp.cap + ix];
                    if (wd.w && r + wd.z <= max_rsi) {
                        if (lane == 0) centry[wv / sp.wpc] = ChunkEntry{good, r, 1u, 0u};
                        good = (uint64_t)wd.x | ((uint64_t)wd.y << 32);
                        r += wd.z;
                        hopped = true;
                        continue;
                    }
                }
                const uint32_t xc = rec.y >> 24, xb = rec.y & 0xFFFFFFu, t = rec.x;
                if (xc && r + xc <= max_rsi && nh < hop_cap) {
                    if (lane == 0) hops[nh] = IdxHop{good, r, xc, 0u};
                    nh++;
                    good += xb;
                    r += xc;
                } else if (t) {
                    if (lane == 0) rsi_off[r] = good;
                    good += t;
                    r++;
                } else {
                    break;
                }
                hopped = true;
            }
            while (tabs.T && r < max_rsi && good >= tabs.lo && good < tabs.hi && good < end_bit) {
                const uint64_t i = good - tabs.lo;
                const uint32_t xc = tabs.Xc[i], xb = tabs.Xb[i], t = tabs.T[i];
                if (xc && r + xc <= max_rsi && nh < hop_cap) {
                    if (lane == 0) hops[nh] = IdxHop{good, r, xc, 0u};
                    nh++;
                    good += xb;
                    r += xc;
                } else if (t) {
                    if (lane == 0) rsi_off[r] = good;
                    good += t;
                    r++;
                    if (c.flags & F_PAD_RSI) good = (good + 7u) & ~7ull;
                } else {
                    break;
                }
                hopped = true;
            }
            if (r >= max_rsi) break;
            if (carry && !last && good >= (sp.bitmap ? sp.hi : (ht.h16 ? ht.hi : tabs.hi))) {   // the next table chunk continues from here
                if (lane == 0) {
                    carry->good = good;
                    carry->r = r;
                    carry->active = 1;
                    carry->n_hops = nh;
                    carry->b = 0;
                    carry->cur_start = good;
                }
                return;
            }
            if (lane == 0) rsi_off[r] = good;
            cur_start = good;
        }
        // Hop tables (long RSIs): 256 / 64 / 16 coded data sets per lookup while that many blocks are left in
        // the RSI; the coded data set with the reference sample and rest-of-segment runs are parsed below.
        if (ht.h16 && !(pp && b == 0)) {
            bool rsi_done = false;
            while (good >= ht.lo && good < ht.hi && good < end_bit) {
                const uint64_t i = good - ht.lo;
                const uint32_t left = c.rsi - b;
                const uint32_t n0 = ht.n0;
                const uint32_t e256 = (ht.h256 && left >= 16u * n0) ? ht.h256[i] : 0u;
                const uint32_t e64 = left >= 4u * n0 ? ht.h64[i] : 0u;
                const uint32_t e16 = left >= n0 ? ht.h16[i] : 0u;
                if (e256 && 16u * n0 + (e256 >> 24) <= left) {
                    good += e256 & 0xFFFFFFu;
                    b += 16u * n0 + (e256 >> 24);
                } else if (e64 && 4u * n0 + (e64 >> 24) <= left) {
                    good += e64 & 0xFFFFFFu;
                    b += 4u * n0 + (e64 >> 24);
                } else if (e16 && n0 + (e16 >> 13) <= left) {
                    good += e16 & kHopBitsMask;
                    b += n0 + (e16 >> 13);
                } else {
                    break;
                }
                hopped = true;
                stale = true;
                if (b >= c.rsi) {
                    b = 0;
                    r++;
                    rsi_done = true;
                    break;
                }
            }
            if (rsi_done) continue;              // next RSI: back to the top (offset table, table hops)
            if (carry && !last && good >= ht.hi) {   // the chunk ends inside this RSI: the next one goes on from here
                if (lane == 0) {
                    carry->good = good;
                    carry->r = r;
                    carry->active = 1;
                    carry->n_hops = nh;
                    carry->b = b;
                    carry->cur_start = cur_start;
                }
                return;
            }
        }
        // keep the whole next CDS (and the readers' look-ahead) inside the LDS window
        if ((good >> 5) + (coop ? 66u : maxw + 2u) > base + kIdxWindowWords) {
            __syncthreads();
            refill(good >> 5);
            br.init(LdsWindowFetch{win, base}, end_bit, good);
            if (coop) load_regs(good >> 5);
            stale = false;
        } else if (hopped || stale) {
            br.init(LdsWindowFetch{win, base}, end_bit, good);
            stale = false;
        }
        const uint32_t ref = (pp && b == 0) ? 1u : 0u;
        uint32_t nblk = 1;
        bool done = false;
        if (coop) {
            uint32_t rel = (uint32_t)(good - wbase * 32u);
            if (good < wbase * 32u || rel + maxbits + 64u > 2048u) {     // slide the register window
                load_regs(good >> 5);
                rel = (uint32_t)(good & 31u);
            }
            const uint32_t h = peek(rel);
            const uint32_t id = h >> (32u - c.id_len);
            uint32_t q = rel + c.id_len;
            if (id == 0) {
                const uint32_t sel = (h >> (31u - c.id_len)) & 1u;
                q += 1u + ref * c.bps;
                if (sel) {
                    q = skip_ones(q, c.bs / 2);
                } else {
                    const uint32_t e = skip_ones(q, 1);
                    if (e != 0xFFFFFFFFu) {
                        uint32_t nz = e - q;                 // fs + 1
                        if (nz == 5) {
                            const uint32_t left_rsi = c.rsi - b, left_seg = 64u - (b % 64u);
                            nz = left_rsi < left_seg ? left_rsi : left_seg;
                        } else if (nz > 5) {
                            nz--;
                        }
                        if (nz > c.rsi - b) status = DEC_DATA_ERROR;
                        nblk = nz;
                    }
                    q = e;
                }
            } else if (id == idmax) {
                q += c.bs * c.bps;
            } else {
                q += ref * c.bps;
                q = skip_ones(q, c.bs - ref);
                if (q != 0xFFFFFFFFu) q += (c.bs - ref) * (id - 1u);
            }
            if (q != 0xFFFFFFFFu) {
                const uint64_t end = wbase * 32u + q;
                if (status == DEC_DATA_ERROR && end <= end_bit) break;
                if (end > end_bit) {
                    status = DEC_NEED_INPUT;
                    break;
                }
                good = end;
                done = true;
            }
        }
        if (!done) {          // large blocks, or a code reaching beyond the register window
            if (coop) br.init(LdsWindowFetch{win, base}, end_bit, good);
            const uint32_t st = skip_cds(br, c, ref, b, nblk);
            if (st != DEC_OK) {
                status = st;
                break;
            }
            good = br.pos;
        }
        b += nblk;
        if (b >= c.rsi) {
            b = 0;
            r++;
        }
    }
    if (lane == 0 && carry) {
        carry->active = 0;
        carry->n_hops = nh;
    }
    if (lane == 0) {
        // streaming callers: where the trailing partial RSI began, in a slot of its own behind the table
        if (tail_slot && !chunk_off) rsi_off[max_rsi] = cur_start;
        res->n_rsi = r;
        res->tail_blocks = b;
        res->end_bit = good;
        if (!chunk_off) res->pad = status == DEC_NEED_INPUT ? 1u : 0u;
        if (chunk_off) {               // per-stream records are written in full (no init kernel)
            res->status = status == DEC_DATA_ERROR ? DEC_DATA_ERROR : DEC_OK;
            res->pad = 0;
            res->bad_rsi = status == DEC_DATA_ERROR ? r : ~0ull;
        } else if (status == DEC_DATA_ERROR) {
            res->status = DEC_DATA_ERROR;
            res->bad_rsi = r;
        }
    }
}



struct SpecGeom {
    bool ok;
    uint32_t core, look, threads;
    size_t lds;
    uint64_t chunk_bits;     // bit positions tabulated per k_spec launch (multiple of core)
};

constexpr size_t kSpecLdsMax = 160u * 1024u;
constexpr uint32_t kSpecWMax = 24576;                    // largest window considered (bits)
constexpr uint64_t kSpecChunkBits = 1ull << 25;          // 4 MiB of stream per table chunk

size_t spec_lds_bytes(uint32_t core, uint32_t look)
{
    const uint32_t W = core + look, nw = W / 32;
    return (size_t)(nw + 2) * 4 + (size_t)(nw + 2) * 2 * 2 + (size_t)W * 2 * 3 + (size_t)core * 2;
}

// The tables pay off when whole RSIs fit the look-ahead of a window.  `rsi_bits_hint` is the
// caller's estimate of the average coded RSI (stream bits / expected RSIs; 0 = unknown): the
// look-ahead is twice that, the rest of the LDS budget is the core.  RSIs longer than the look-ahead
// are left to the serial walk, so a wrong hint costs speed only.  Small inputs get small cores so
// that the windows still fill the chip.
SpecGeom spec_geom(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint)
{
    SpecGeom g{};
    const uint64_t worst = (uint64_t)c.rsi * (c.id_len + (uint64_t)c.bs * c.bps) + c.bps + 64;
    if (total_bits < 4096) return g;
    uint64_t look;
    if (rsi_bits_hint) {
        if (rsi_bits_hint + rsi_bits_hint / 4 + 256 > kSpecWMax - 2048) return g;
        look = 2 * rsi_bits_hint + 1024;
    } else {
        if (worst > 65000) return g;
        look = 12288;
    }
    if (look > worst) look = worst;
    if (look > kSpecWMax - 2048) look = kSpecWMax - 2048;
    if (look < 2048) look = 2048;
    g.look = (uint32_t)((look + 31) & ~31ull);
    // about one window per CU for small inputs, and windows of at least three RSIs (a hop of the
    // walker costs a memory round trip: it should cover several RSIs)
    uint64_t core = (total_bits / 256 + 1023) & ~1023ull;
    uint64_t three = 3 * rsi_bits_hint;
    if (three > total_bits / 64) three = total_bits / 64;      // (tiny inputs: keep 64 windows at least)
    if (core < three) core = (three + 1023) & ~1023ull;
    if (core < 2048) core = 2048;
    if (core > 16384) core = 16384;
    if (core > kSpecWMax - g.look) core = (kSpecWMax - g.look) & ~1023ull;
    while (core >= 2048 && spec_lds_bytes((uint32_t)core, g.look) > kSpecLdsMax) core -= 1024;
    if (core < 1024) return g;
    g.core = (uint32_t)core;
    g.lds = spec_lds_bytes(g.core, g.look);
    if (g.lds > kSpecLdsMax) return g;
    const uint32_t W = g.core + g.look;
    g.threads = W >= 16384 ? 1024 : (W >= 8192 ? 512 : 256);
    g.chunk_bits = (kSpecChunkBits / g.core) * g.core;
    g.ok = true;
    return g;
}

void allow_big_lds()
{
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_spec),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSpecLdsMax);
    });
}

}  // namespace

namespace {

// One set of tables (+ the walker's hop list) for a chunk of `entries` bit positions.
struct TableSet {
    uint16_t *T, *Xb;
    uint8_t *Xc;
    IdxHop *hops;
};

size_t table_set_bytes(uint64_t entries, uint64_t nwin)
{
    return (size_t)((entries * 5 + 15) / 16 * 16 + (2 * nwin + 16) * sizeof(IdxHop));
}

TableSet table_set_at(uint8_t *p, uint64_t entries)
{
    TableSet t;
    t.T = reinterpret_cast<uint16_t *>(p);
    t.Xb = t.T + entries;
    t.Xc = reinterpret_cast<uint8_t *>(t.Xb + entries);
    t.hops = reinterpret_cast<IdxHop *>(p + (entries * 5 + 15) / 16 * 16);
    return t;
}

}  // namespace

namespace {

// ---- sparse path: geometry and workspace ------------------------------------------------------------
struct Sparse2Plan {
    bool ok;
    Spec2Geom g;
    size_t lds;
    uint32_t nwin_max;        // windows per super-chunk (one k_spec2 launch)
    uint32_t wpc;             // windows per chunk of the wide walker
    uint32_t nchunk_max;
    // byte offsets inside the workspace (behind the 64-byte carry record)
    size_t o_bitmap, o_pre, o_rec, o_cpos, o_ccnt, o_wide, o_centry, o_hops, o_rhops, o_nhops, bytes;
};

constexpr uint32_t kS2WindowBits = 65536;      // lead-in + core + look-ahead (16-bit positions in LDS)
constexpr uint32_t kS2Lead = 4096;
constexpr uint32_t kS2SuperWindows = 4096;     // windows per launch: bounds the table workspace (~60 KB each)

// The sparse speculation pays off where the coded data sets are short and unary-dominated, so that
// the sync chains (aec_spec2.h) fall onto the true chain within a few codes: low-entropy data, a few
// bits per sample (measured in tests/emul: BASELINE configs 2 and 5 -- no RSI start missed with a
// 4-kbit lead-in; 32-bit data with 8-bit fields -- chains take tens of kbit to merge, not used there).
// It needs an estimate of the coded RSI size (look-ahead) and whole RSIs inside a window.
Sparse2Plan sparse2_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint)
{
    Sparse2Plan p{};
    static const bool off = getenv("AEC_IDX_DENSE") != nullptr;          // A/B switch for measurements
    if (off || (c.flags & F_PAD_RSI) || !rsi_bits_hint || total_bits < 16384) return p;
    const uint64_t samples = (uint64_t)c.rsi * c.bs;
    if (rsi_bits_hint * 2 > samples * 9) return p;                       // more than 4.5 bits per sample
    uint64_t look = (2 * rsi_bits_hint + 1024 + 31) & ~31ull;
    if (look < 4096) look = 4096;
    static const char *e_win = getenv("AEC_S2_WINDOW");
    uint32_t wbits = e_win ? (uint32_t)atoi(e_win) : kS2WindowBits;
    if (wbits < 16384 || wbits > kS2WindowBits) wbits = kS2WindowBits;
    if (look + kS2Lead + 8192 > wbits) wbits = kS2WindowBits;            // long RSIs: the largest window
    if (look > kS2WindowBits - kS2Lead - 16384) return p;
    uint64_t core = (wbits - kS2Lead - look) & ~1023ull;
    // small inputs: about one window per CU
    const uint64_t want = ((total_bits / 256 + 1023) & ~1023ull);
    if (core > want) core = want < 8192 ? 8192 : want;
    p.g.lead = kS2Lead;
    p.g.core = (uint32_t)core;
    p.g.look = (uint32_t)look;
    static const char *e_stride = getenv("AEC_S2_STRIDE"), *e_burn = getenv("AEC_S2_BURN");
    p.g.stride = e_stride ? (uint32_t)atoi(e_stride) : 64u;
    p.g.burn = e_burn ? (uint32_t)atoi(e_burn) : 24u;
    static const char *e_budget = getenv("AEC_S2_BUDGET");
    p.g.budget = e_budget ? (uint32_t)atoi(e_budget) : 0xFFFFFFFFu;
    const uint32_t W = p.g.lead + p.g.core + p.g.look, nw = W / 32;
    static const char *e_capdiv = getenv("AEC_S2_CAPDIV");
    const uint32_t capdiv = e_capdiv ? (uint32_t)atoi(e_capdiv) : 8u;
    p.g.cap_lds = (W / (capdiv ? capdiv : 8u) + 63) & ~63u;
    p.g.cap_core = (p.g.core / 8 + 63) & ~63u;
    p.lds = (size_t)(nw + 2) * 4 + (size_t)nw * 4 + (size_t)(nw + 2) * 2 * 3 + (size_t)p.g.cap_lds * 2 * 5 + 64;
    if (p.lds > 156 * 1024) return p;
    const uint64_t nwin_total = (total_bits + core + core - 1) / core;    // (+ one: the range starts on a core boundary)
    p.nwin_max = (uint32_t)(nwin_total < kS2SuperWindows ? nwin_total : kS2SuperWindows);
    p.wpc = p.nwin_max >= 2048 ? 256u : (p.nwin_max >= 64 ? p.nwin_max / 8 : p.nwin_max);
    if (p.wpc == 0) p.wpc = 1;
    p.nchunk_max = (p.nwin_max + p.wpc - 1) / p.wpc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t words = (size_t)p.nwin_max * (p.g.core / 32);
    size_t o = 64;
    p.o_bitmap = o; o = up(o + words * 4);
    p.o_pre = o;    o = up(o + words * 2);
    p.o_rec = o;    o = up(o + (size_t)p.nwin_max * p.g.cap_core * sizeof(uint2));
    p.o_cpos = o;   o = up(o + (size_t)p.nwin_max * p.g.cap_core * 2);
    p.o_ccnt = o;   o = up(o + (size_t)p.nwin_max * 4);
    p.o_wide = o;   o = up(o + (size_t)p.nchunk_max * p.g.cap_core * sizeof(uint4));
    p.o_centry = o; o = up(o + (size_t)p.nchunk_max * sizeof(ChunkEntry));
    p.o_hops = o;   o = up(o + ((size_t)p.nwin_max * 2 + 16) * sizeof(IdxHop));
    p.o_rhops = o;  o = up(o + (size_t)p.nchunk_max * p.wpc * 2 * sizeof(IdxHop));
    p.o_nhops = o;  o = up(o + (size_t)p.nchunk_max * 4);
    p.bytes = o;
    p.ok = true;
    return p;
}

// AEC_S2_PROF=1: phase stamps of k_spec2 (diagnostics; printed by the host at exit of the first launch)
unsigned long long *spec2_prof_buffer(uint32_t nwin, bool reset = true)
{
    static const bool on = getenv("AEC_S2_PROF") != nullptr;
    static unsigned long long *buf = nullptr;
    if (!on) return nullptr;
    if (!buf) (void)hipMalloc(reinterpret_cast<void **>(&buf), (size_t)kS2SuperWindows * 8 * sizeof(unsigned long long));
    if (reset) (void)hipMemset(buf, 0, (size_t)kS2SuperWindows * 8 * sizeof(unsigned long long));
    (void)nwin;
    return buf;
}

void spec2_prof_report(uint32_t nwin, hipStream_t st)
{
    unsigned long long *buf = spec2_prof_buffer(nwin, false);
    if (!buf) return;
    static int reports = 0;
    if (reports++ >= 2) return;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nwin * 8);
    (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
    double d[6] = {0, 0, 0, 0, 0, 0};
    uint32_t n = 0;
    for (uint32_t w = 0; w + 1 < nwin; w++) {
        if (!h[(size_t)w * 8 + 6]) continue;
        for (int k = 0; k < 6; k++) d[k] += (double)(h[(size_t)w * 8 + k + 1] - h[(size_t)w * 8 + k]);
        n++;
    }
    fprintf(stderr, "k_spec2 phases (shader-clock ticks per window, %u windows): load+rank %.0f | chains %.0f | "
            "prefix+cand nxt %.0f | hop4+hop16 %.0f | units %.0f | chain+write %.0f\n", n, d[0] / n, d[1] / n,
            d[2] / n, d[3] / n, d[4] / n, d[5] / n);
}

void allow_big_lds2()
{
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [] {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_spec2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                156 * 1024) != hipSuccess)
            (void)hipGetLastError();
    });
}

// Index pass over the sparse tables, super-chunk by super-chunk: speculation (all CUs), wide walker
// (every candidate of every chunk's first window), the walk (one wavefront: one lookup per chunk where
// the wide table resolves it, per window or per coded data set where not), then the true chain through
// the skipped chunks and the RSI starts inside all hops.
void launch_index_sparse(const Cfg &c, const Sparse2Plan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                         uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                         uint8_t *base, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot)
{
    allow_big_lds2();
    IdxCarry *carry = reinterpret_cast<IdxCarry *>(base);
    const uint64_t lo0 = start_bit / p.g.core * p.g.core;
    const uint64_t span = (uint64_t)p.nwin_max * p.g.core;
    for (uint64_t lo = lo0; lo < end_bit; lo += span) {
        const uint64_t bits = end_bit - lo < span ? end_bit - lo : span;
        const uint32_t nwin = (uint32_t)((bits + p.g.core - 1) / p.g.core);
        const uint32_t nchunks = (nwin + p.wpc - 1) / p.wpc;
        const bool first = lo == lo0, last = lo + span >= end_bit;
        SparseTables t;
        t.bitmap = reinterpret_cast<const uint32_t *>(base + p.o_bitmap);
        t.pre = reinterpret_cast<const uint16_t *>(base + p.o_pre);
        t.rec = reinterpret_cast<const uint2 *>(base + p.o_rec);
        t.cpos = reinterpret_cast<const uint16_t *>(base + p.o_cpos);
        t.ccnt = reinterpret_cast<const uint32_t *>(base + p.o_ccnt);
        t.lo = lo;
        t.hi = lo + (uint64_t)nwin * p.g.core;
        t.core = p.g.core;
        t.cap = p.g.cap_core;
        t.wide = reinterpret_cast<const uint4 *>(base + p.o_wide);
        t.wpc = p.wpc;
        ChunkEntry *centry = reinterpret_cast<ChunkEntry *>(base + p.o_centry);
        IdxHop *hops = reinterpret_cast<IdxHop *>(base + p.o_hops);
        IdxHop *rhops = reinterpret_cast<IdxHop *>(base + p.o_rhops);
        uint32_t *nhops = reinterpret_cast<uint32_t *>(base + p.o_nhops);
        const uint32_t hop_cap = 2 * nwin + 8;
        (void)hipMemsetAsync(base + p.o_wide, 0, (size_t)nchunks * p.g.cap_core * sizeof(uint4), st);
        (void)hipMemsetAsync(centry, 0, (size_t)nchunks * sizeof(ChunkEntry), st);
        hipLaunchKernelGGL(k_spec2, dim3(nwin), dim3(1024), p.lds, st, c, words, nwords, end_bit, lo, start_bit, p.g,
                           const_cast<uint32_t *>(t.bitmap), const_cast<uint16_t *>(t.pre), const_cast<uint2 *>(t.rec),
                           const_cast<uint16_t *>(t.cpos), const_cast<uint32_t *>(t.ccnt), spec2_prof_buffer(nwin));
        spec2_prof_report(nwin, st);
        hipLaunchKernelGGL(k_wide, dim3((p.g.cap_core + 255) / 256, nchunks), dim3(256), 0, st, t, nwin, end_bit,
                           const_cast<uint4 *>(t.wide));
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi,
                           d_res, (const uint64_t *)nullptr, IdxTables{}, hops, hop_cap, carry, first ? 1u : 0u,
                           last ? 1u : 0u, start_block, rsi_start, tail_slot, t, centry, HopTables{});
        hipLaunchKernelGGL(k_rewalk, dim3((nchunks + 63) / 64), dim3(64), 0, st, t, nwin, nchunks, end_bit, centry, rhops,
                           nhops);
        hipLaunchKernelGGL(k_expand2, dim3((hop_cap + 255) / 256), dim3(256), 0, st, t, carry, hops,
                           (const uint32_t *)nullptr, 0u, 0u, d_rsi_off);
        hipLaunchKernelGGL(k_expand2, dim3((nchunks * p.wpc * 2 + 255) / 256), dim3(256), 0, st, t, carry, rhops, nhops,
                           nchunks, p.wpc * 2, d_rsi_off);
    }
}

}  // namespace

namespace {

// ---- hop path: geometry and launches ------------------------------------------------------------------
struct HopPlan {
    bool ok;
    uint32_t core, look, n0;
    size_t lds;
    uint64_t chunk_bits;      // positions tabulated per launch (multiple of core)
    size_t set_bytes;         // one table set: h16 + h64 + h256 for chunk_bits positions
};

constexpr uint32_t kHopWindowBits = 32768;
constexpr uint64_t kHopChunkBits = 1ull << 25;       // 4 MiB of stream per table chunk (10 bytes per bit)

// Used where neither the sparse nor the RSI tables apply: RSIs longer than any window (BASELINE config 3:
// a megabit coded per RSI).  The look-ahead must hold the coded data sets of a base entry (16, or 4).
HopPlan hop_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint)
{
    HopPlan p{};
    static const bool off = getenv("AEC_IDX_NO_HOPS") != nullptr;
    if (off || (c.flags & F_PAD_RSI) || total_bits < 65536 || c.rsi < 64) return p;
    // average coded data set from the hint, else half the uncompressed size
    uint64_t cds = rsi_bits_hint ? rsi_bits_hint / c.rsi : (uint64_t)(c.id_len + c.bs * c.bps) / 2;
    if (cds < 32) cds = 32;
    p.n0 = cds * 16 * 3 / 2 < kHopBitsMask ? 16u : 4u;         // (distance of a base entry: 13 bits)
    uint64_t look = (p.n0 * cds * 3 / 2 + 1023) & ~1023ull;
    if (look < 4096) look = 4096;
    if (look > kHopWindowBits - 8192) return p;
    p.look = (uint32_t)look;
    p.core = (uint32_t)((kHopWindowBits - look) & ~1023ull);
    const uint32_t W = p.core + p.look, nw = W / 32;
    p.lds = (size_t)(nw + 2) * 4 + (size_t)(nw + 2) * 2 * 2 + (size_t)W * 2 * 2;
    if (p.lds > 156 * 1024) return p;
    p.chunk_bits = (kHopChunkBits / p.core) * p.core;
    p.set_bytes = (size_t)((p.chunk_bits * 10 + 255) & ~255ull);
    p.ok = true;
    return p;
}

void allow_big_lds_hops()
{
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [] {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_hops), hipFuncAttributeMaxDynamicSharedMemorySize,
                                156 * 1024) != hipSuccess)
            (void)hipGetLastError();
    });
}

void launch_index_hops(const Cfg &c, const HopPlan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                       uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                       uint8_t *base, const IdxSide *side, uint32_t start_block, uint64_t rsi_start,
                       uint32_t tail_slot)
{
    allow_big_lds_hops();
    IdxCarry *carry = reinterpret_cast<IdxCarry *>(base);
    const uint64_t lo0 = start_bit / p.core * p.core;
    const bool multi = end_bit - lo0 > p.chunk_bits;
    const bool piped = multi && side && side->stream;
    hipStream_t wst = piped ? side->stream : st;
    if (piped)
        for (int b = 0; b < 2; b++) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
    uint32_t i = 0;
    for (uint64_t lo = lo0; lo < end_bit; lo += p.chunk_bits, i++) {
        uint64_t bits = end_bit - lo;
        if (bits > p.chunk_bits) bits = p.chunk_bits;
        const uint32_t nwin = (uint32_t)((bits + p.core - 1) / p.core), b = i & 1u;
        const uint64_t n = (uint64_t)nwin * p.core;
        const bool first = lo == lo0, last = lo + p.chunk_bits >= end_bit;
        uint8_t *set = base + 64 + (multi ? (size_t)b * p.set_bytes : 0);
        uint16_t *h16 = reinterpret_cast<uint16_t *>(set);
        uint32_t *h64 = reinterpret_cast<uint32_t *>(set + ((p.chunk_bits * 2 + 255) & ~255ull));
        uint32_t *h256 = h64 + p.chunk_bits;
        if (piped && i >= 2) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
        hipLaunchKernelGGL(k_hops, dim3(nwin), dim3(1024), p.lds, st, c, words, nwords, end_bit, lo, p.core, p.look, h16,
                           p.n0);
        hipLaunchKernelGGL((k_hop_compose<true>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                           (const void *)h16, h64, n);
        hipLaunchKernelGGL((k_hop_compose<false>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                           (const void *)h64, h256, n);
        if (piped) {
            (void)hipEventRecord(side->spec_done[b], st);
            (void)hipStreamWaitEvent(wst, side->spec_done[b], 0);
        }
        const HopTables ht{h16, h64, h256, lo, lo + n, p.n0};
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, wst, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi,
                           d_res, (const uint64_t *)nullptr, IdxTables{}, (IdxHop *)nullptr, 0u, carry,
                           first ? 1u : 0u, last ? 1u : 0u, start_block, rsi_start, tail_slot, SparseTables{},
                           (ChunkEntry *)nullptr, ht);
        if (piped) (void)hipEventRecord(side->walk_done[b], wst);
    }
    if (piped)
        for (int b = 0; b < 2; b++) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
}

}  // namespace

size_t index_workspace_bytes(const Cfg &c, size_t in_bytes, uint64_t start_bit, uint64_t rsi_bits_hint)
{
    const uint64_t end_bit = (uint64_t)in_bytes * 8;
    if (start_bit >= end_bit) return 0;
    const Sparse2Plan sp = sparse2_plan(c, end_bit - start_bit, rsi_bits_hint);
    if (sp.ok) return sp.bytes;
    const SpecGeom g = spec_geom(c, end_bit - start_bit, rsi_bits_hint);
    if (!g.ok) {
        const HopPlan hp = hop_plan(c, end_bit - start_bit, rsi_bits_hint);
        if (!hp.ok) return 0;
        return 64 + hp.set_bytes * (end_bit - start_bit / hp.core * hp.core > hp.chunk_bits ? 2 : 1);
    }
    const uint64_t lo = start_bit / g.core * g.core;
    uint64_t span = end_bit - lo;
    const bool multi = span > g.chunk_bits;
    if (multi) span = g.chunk_bits;
    const uint64_t nwin = (span + g.core - 1) / g.core;
    return 64 + table_set_bytes(nwin * g.core, nwin) * (multi ? 2 : 1);   // two sets: spec(i+1) beside walk(i)
}

void launch_index(const Cfg &c, const uint8_t *d_in, size_t in_bytes, uint64_t start_bit,
                  uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                  void *d_ws, size_t ws_bytes, uint64_t rsi_bits_hint, const IdxSide *side,
                  uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot)
{
    const uint32_t *words = reinterpret_cast<const uint32_t *>(d_in);
    const uint64_t nwords = (in_bytes + 3) / 4, end_bit = (uint64_t)in_bytes * 8;
    const size_t need = index_workspace_bytes(c, in_bytes, start_bit, rsi_bits_hint);
    if (!need || !d_ws || ws_bytes < need) {           // serial walk only
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off,
                           max_rsi, d_res, (const uint64_t *)nullptr, IdxTables{}, (IdxHop *)nullptr, 0u,
                           (IdxCarry *)nullptr, 1u, 1u, start_block, rsi_start, tail_slot, SparseTables{}, (ChunkEntry *)nullptr,
                           HopTables{});
        return;
    }
    const Sparse2Plan sp = sparse2_plan(c, end_bit - start_bit, rsi_bits_hint);
    if (sp.ok) {
        launch_index_sparse(c, sp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                            static_cast<uint8_t *>(d_ws), start_block, rsi_start, tail_slot);
        return;
    }
    const SpecGeom g = spec_geom(c, end_bit - start_bit, rsi_bits_hint);
    if (!g.ok) {
        launch_index_hops(c, hop_plan(c, end_bit - start_bit, rsi_bits_hint), words, nwords, end_bit, start_bit,
                          d_rsi_off, max_rsi, d_res, st, static_cast<uint8_t *>(d_ws), side, start_block, rsi_start,
                          tail_slot);
        return;
    }
    allow_big_lds();
    const uint64_t lo0 = start_bit / g.core * g.core;
    uint64_t span = end_bit - lo0;
    const bool multi = span > g.chunk_bits;
    if (multi) span = g.chunk_bits;
    const uint64_t nwin_max = (span + g.core - 1) / g.core, entries = nwin_max * g.core;
    uint8_t *base = static_cast<uint8_t *>(d_ws);
    IdxCarry *carry = reinterpret_cast<IdxCarry *>(base);
    const TableSet set[2] = {table_set_at(base + 64, entries),
                             table_set_at(base + 64 + (multi ? table_set_bytes(entries, nwin_max) : 0), entries)};
    const uint32_t hop_cap = (uint32_t)(2 * nwin_max + 8);
    // With several chunks the walk over chunk i (one wavefront) runs on a side stream beside the
    // speculation over chunk i + 1 (all CUs) on the caller's stream; events order the two table sets.
    const bool piped = multi && side && side->stream;
    hipStream_t wst = piped ? side->stream : st;
    if (piped)                                           // sets may still be read by an earlier call's walker
        for (int b = 0; b < 2; b++) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
    uint32_t i = 0;
    for (uint64_t lo = lo0; lo < end_bit; lo += g.chunk_bits, i++) {
        uint64_t bits = end_bit - lo;
        if (bits > g.chunk_bits) bits = g.chunk_bits;
        const uint32_t nwin = (uint32_t)((bits + g.core - 1) / g.core), b = i & 1u;
        const bool first = lo == lo0, last = lo + g.chunk_bits >= end_bit;
        const TableSet &t = set[b];
        const IdxTables tabs{t.T, t.Xb, t.Xc, lo, lo + (uint64_t)nwin * g.core};
        if (piped && i >= 2) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
        hipLaunchKernelGGL(k_spec, dim3(nwin), dim3(g.threads), g.lds, st, c, words, nwords, end_bit, lo, g.core,
                           g.look, t.T, t.Xb, t.Xc);
        if (piped) {
            (void)hipEventRecord(side->spec_done[b], st);
            (void)hipStreamWaitEvent(wst, side->spec_done[b], 0);
        }
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, wst, c, words, nwords, end_bit, start_bit, d_rsi_off,
                           max_rsi, d_res, (const uint64_t *)nullptr, tabs, t.hops, hop_cap, carry,
                           first ? 1u : 0u, last ? 1u : 0u, start_block, rsi_start, tail_slot, SparseTables{},
                           (ChunkEntry *)nullptr, HopTables{});
        hipLaunchKernelGGL(k_expand, dim3((hop_cap + 255) / 256), dim3(256), 0, wst, carry, t.hops, tabs,
                           (c.flags & F_PAD_RSI) ? 1u : 0u, d_rsi_off);
        if (piped) (void)hipEventRecord(side->walk_done[b], wst);
    }
    if (piped)
        for (int b = 0; b < 2; b++) (void)hipStreamWaitEvent(st, side->walk_done[b], 0);
}

void launch_index_batch(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_chunk_off,
                        uint64_t n_chunks, uint64_t rsi_per_chunk, uint64_t *d_rsi_off, DecResult *d_res,
                        hipStream_t st)
{
    if (n_chunks == 0) return;
    hipLaunchKernelGGL(k_index, dim3((uint32_t)n_chunks), dim3(64), 0, st, c,
                       reinterpret_cast<const uint32_t *>(d_in), (uint64_t)((in_bytes + 3) / 4),
                       (uint64_t)in_bytes * 8, (uint64_t)0, d_rsi_off, rsi_per_chunk, d_res, d_chunk_off,
                       IdxTables{}, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u, 1u, 0u, (uint64_t)0, 0u, SparseTables{},
                       (ChunkEntry *)nullptr, HopTables{});
}

}  // namespace aec
