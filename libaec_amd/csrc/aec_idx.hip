// aec_idx.hip -- RSI index pass for streams that arrive without an offset table (gfx950).
//
// A coded data set can only be located by parsing its predecessor (the stream has no lengths,
// reference src/decode.c:402-421), so finding where the RSIs start is a serial walk in the
// reference.  Here it is split into lane-parallel passes over the stream (aec_trunk.h has the
// arithmetic and the reasoning) and a walk that only hops over their results:
//   k_trunk       one lane per region: the self-synchronising parse WITHOUT reference samples, burnt in
//                 over `lead` bits, records every coded-data-set boundary it visits (the NODES of the trunk)
//                 window by window; repair passes walk a region again from where its left neighbour ended
//   k_trunk_scan  numbers nodes and blocks along the trunk, counts seams and rest-of-segment runs
//   k_hyp_walk    every node is tried as an RSI start: the coded data set with the reference sample, then
//                 on-demand parses until the walk stands on the trunk again (or has completed up to 63
//                 RSIs without meeting it)
//   k_hyp_land    the rest of the RSI as ONE search in the block numbering of the trunk
//   k_hyp_chain   records chained until they leave their window
//   k_twide        every node of the first window of every chunk (256 windows) chases the chain through the
//                 chunk, so that the walk needs one lookup per chunk
//   k_index       the WALK, one wavefront: hops over those tables; an RSI they did not resolve (cut by the
//                 end of the stream, malformed, a seam) is walked CDS by CDS: the stream is served from an
//                 LDS window, unary parts are skipped cooperatively (popcount per lane + DPP scan)
//   k_trewalk, k_texpand  fill in the RSI starts inside the hops the walk took
// Every record is the exact result of the reference's walk from its node; the walk reads records at true
// RSI starts only, so results never depend on the speculation.  The batch form runs one serial walker per
// independent chunk.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "aec_kernels.h"
#include "aec_lane.h"
#include "aec_spec.h"
#include "aec_spec2.h"
#include "aec_trunk.h"
#include "aec_coop.h"
#include "aec_stretch.h"
#include "aec_small.h"
#include "aec_tune.h"

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

// What the walkers read of the trunk tables of one span of the stream (aec_trunk.h: TrTables): per window k
// (core bits each, the first one at bit lo) a bitmap over its bit positions, per bitmap word the number of
// nodes in front of it (inside the window), and the nodes' records, stored back to back.
struct TwTables {
    const uint32_t *bitmap;    // bit (31 - i % 32) of word i / 32 <=> position lo + i is a node
    const uint16_t *pre;       // per bitmap word: nodes of ITS window in front of the word
    const uint32_t *nbase;     // [window]: nodes in front of the window
    const uint32_t *ccnt;      // [window]: nodes
    const TrRec *rec;          // [nbase[window] + index]
    const uint16_t *cpos;      // [nbase[window] + index]: position inside the window
    uint64_t lo, hi;           // bit range whose nodes have records
    uint32_t core;             // bits per window (multiple of 32)
    // wide walker: per chunk of wpc windows, what every node of the chunk's first wfirst windows leads to (index =
    // the node's number among those: record index - nbase[first window of the chunk])
    const uint4 *wide;         // [chunk * wcap + index]: {exit lo, exit hi, RSIs, 1 = resolved}
    uint32_t wpc, wcap, wfirst;
    uint32_t nrec;             // records there is room for (indices beyond it are never read)
    const uint32_t *skip_if;   // non-null and *skip_if != 0: another scheme has delivered the index, the kernels return
};

// record of the node at absolute bit p (false: p is not a node)
__device__ __forceinline__ bool tw_lookup(const TwTables &t, uint64_t p, TrRec &rec, uint32_t &window,
                                              uint32_t &index)
{
    const uint64_t i = p - t.lo;
    const uint32_t word = t.bitmap[i >> 5], sh = (uint32_t)(i & 31u);
    if (!((word >> (31u - sh)) & 1u)) return false;
    window = (uint32_t)(i / t.core);
    index = (uint32_t)t.pre[i >> 5] + (sh ? (uint32_t)__popc(word >> (32u - sh)) : 0u);
    rec = t.rec[t.nbase[window] + index];
    return true;
}

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
    const uint32_t *skip_if = nullptr;   // != 0 there: a scheme in front has delivered the stream; the kernels return at once
};

// The DENSE tables (launch_index_sparse: windows with room for a candidate per four bits, built only where a window of
// the ordinary tables gave up): a window that was not built has a count of 0 and its part of the bitmap is not written.
__device__ __forceinline__ bool dense_lookup(const SparseTables &t, uint64_t p, uint2 &rec)
{
    const uint64_t i = p - t.lo;
    const uint32_t window = (uint32_t)(i / t.core);
    if (!t.ccnt[window]) return false;
    const uint32_t word = t.bitmap[i >> 5], sh = (uint32_t)(i & 31u);
    if (!((word >> (31u - sh)) & 1u)) return false;
    const uint32_t index = (uint32_t)t.pre[i >> 5] + (sh ? (uint32_t)__popc(word >> (32u - sh)) : 0u);
    rec = t.rec[(uint64_t)window * t.cap + index];
    return true;
}

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

struct ChunkEntry {        // where the true chain enters a chunk the walker skipped over the wide table
    uint64_t pos, r;
    uint32_t valid, pad;
};

struct IdxHop {            // a hop the walker took: k_texpand writes the RSI starts inside it
    uint64_t pos, r;
    uint32_t cnt, pad;
};

struct IdxCarry {          // walker state between the spans of one stream
    uint64_t good, r;
    uint32_t active, n_hops;
    uint32_t n_serial, n_lookups;     // statistics: RSIs walked coded data set by coded data set, table hops taken
    uint64_t r_prev;                  // RSIs in front of the span the walker took last (k_seg_starts: its RSIs are
                                      // [r_prev, r) while the walk goes on, [r_prev, what the result record says) else)
};
static_assert(sizeof(IdxCarry) <= 48, "the carry record shares 64 bytes with the pool counter at offset 48");

// ---- trunk (aec_trunk.h) -------------------------------------------------------------------------------
// mode 0: first pass (burn-in + count), 1: repair pass (count), 2: fill (one lane per WINDOW)
__global__ void __launch_bounds__(64)
k_trunk(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, const uint64_t *exit_prev, uint64_t *exit_out,
        uint32_t mode)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 2u) {
        if (i < g.nwin) tr_trunk_window(s, c, g, t, i, t.entry[i], nullptr, TR_FILL);
        return;
    }
    const uint32_t nreg = (g.nwin + g.rw - 1u) / g.rw;
    if (i < nreg) tr_trunk_region(s, c, g, t, i, mode ? exit_prev : nullptr, exit_out, TR_COUNT);
}

// The count passes of the trunk with a WAVEFRONT per region (aec_coop.h) instead of a lane: a region is a serial
// chain of a few thousand coded data sets (burn-in + its windows), and the first pass took as long as one lane
// needs for that from device memory.  Same arithmetic and same tables as tr_trunk_region / tr_trunk_window in
// TR_COUNT mode; the words of bitmap and prefix table between two nodes are written by the lanes side by side.
constexpr uint32_t kTrunkCoopWin = 2048;
template <class CW>
__device__ __forceinline__ uint64_t coop_trunk_window(CW &cw, const Cfg &c, const TrGeom &g, const TrTables &t, uint32_t w,
                                                      uint64_t pos, uint64_t *exit_out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wstart = g.lo + (uint64_t)w * g.L, wend = wstart + g.L;
    const uint32_t nw = g.L / 32u;
    uint32_t *bm = t.bitmap + (uint64_t)w * nw;
    uint16_t *pre = t.pre + (uint64_t)w * nw;
    if (lane == 0) t.entry[w] = pos;
    uint32_t cnt = 0, blocks = 0, ros = 0, wi = 0, wv = 0, wpre = 0;
    bool full = false;
    while (pos != kTrNone && pos < wend) {
        uint32_t nzc;
        const uint32_t len = cw.cds(c, pos, 0u, nzc);
        if (!full && blocks > kTrBpMask - 64u) full = true;
        if (!full) {
            const uint32_t rel = (uint32_t)(pos - wstart), word = rel >> 5;
            if (wi < word) {
                for (uint32_t k = wi + lane; k < word; k += 64u) {
                    bm[k] = k == wi ? wv : 0u;
                    pre[k] = (uint16_t)(k == wi ? wpre : cnt);
                }
                wi = word;
                wv = 0;
                wpre = cnt;
            }
            uint32_t nb = 1;
            if (!len) nb = 0;
            else if (nzc == 5u) ros++;
            else if (nzc) nb = nzc > 5u ? nzc - 1u : nzc;
            wv |= 0x80000000u >> (rel & 31u);
            cnt++;
            blocks += nb;
        }
        pos = len ? pos + len : kTrNone;
    }
    for (uint32_t k = wi + lane; k < nw; k += 64u) {
        bm[k] = k == wi ? wv : 0u;
        pre[k] = (uint16_t)(k == wi ? wpre : cnt);
    }
    if (lane == 0) {
        exit_out[w] = full ? kTrNone : pos;
        t.ccnt[w] = cnt;
        t.nblk[w] = blocks;
        t.nros[w] = ros;
    }
    return pos;
}

// mode 0: first pass (burn-in + count), 1: repair pass (count); one wavefront per region
__global__ void __launch_bounds__(64)
k_trunk_coop(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, const uint64_t *exit_prev, uint64_t *exit_out,
             uint32_t mode)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ __attribute__((aligned(16))) uint32_t lds_w[kTrunkCoopWin];
    const uint32_t nreg = (g.nwin + g.rw - 1u) / g.rw, lane = threadIdx.x;
    CoopCds<kTrunkCoopWin> cw;
    cw.init(s, c, lds_w);
    for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x) {
        const uint32_t w0 = r * g.rw, w1 = w0 + g.rw < g.nwin ? w0 + g.rw : g.nwin;
        const uint64_t rstart = g.lo + (uint64_t)w0 * g.L, rend = g.lo + (uint64_t)w1 * g.L;
        uint64_t pos;
        if (!mode) {
            pos = rstart > g.start_bit + g.lead ? rstart - g.lead : g.start_bit;
            if (rend <= g.start_bit || rstart > s.end_bit) pos = kTrNone;
            if (pos != kTrNone && pos != g.start_bit) {                      // (tr_trunk_region: a run of uncompressed headers)
                const uint64_t q = tr_unc_run(s, c, g.start_bit, rstart);
                if (q != kTrNone) pos = q;
            }
            while (pos != kTrNone && pos < rstart) {                         // burn-in
                uint32_t nzc;
                const uint32_t len = cw.cds(c, pos, 0u, nzc);
                pos = len ? pos + len : kTrNone;
            }
        } else {
            const uint64_t prev = w0 ? exit_prev[w0 - 1u] : kTrNone;
            if (prev == kTrNone || prev == t.entry[w0]) {
                for (uint32_t w = w0 + lane; w < w1; w += 64u) exit_out[w] = exit_prev[w];
                continue;
            }
            pos = prev;
        }
        for (uint32_t w = w0; w < w1; w++) {
            if (mode && w > w0 && pos == t.entry[w]) {       // (back on the chain the region held before: tr_trunk_region)
                for (uint32_t u = w + lane; u < w1; u += 64u) exit_out[u] = exit_prev[u];
                break;
            }
            pos = coop_trunk_window(cw, c, g, t, w, pos, exit_out);
        }
    }
}

// inclusive scan of one 64-bit value per lane over a workgroup of 1024 (16 wavefronts); `total` = sum of all
__device__ __forceinline__ uint64_t block_incl_scan64(uint64_t v, uint64_t *sh16, uint64_t &total)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint64_t up = __shfl_up(v, d);
        if (lane >= d) v += up;
    }
    __syncthreads();                         // (sh16 may still be read from the previous scan)
    if (lane == 63u) sh16[wv] = v;
    __syncthreads();
    if (tid < 16u) {
        uint64_t x = sh16[tid];
        for (uint32_t d = 1; d < 16u; d <<= 1) {
            const uint64_t up = __shfl_up(x, d);
            if (tid >= d) x += up;
        }
        sh16[tid] = x;
    }
    __syncthreads();
    total = sh16[15];
    return v + (wv ? sh16[wv - 1u] : 0u);
}

// One workgroup: exclusive scans over the windows (nodes, blocks, seams, rest-of-segment runs); the same
// rules as tr_scan_serial.  Four consecutive windows per lane and round.
__global__ void __launch_bounds__(1024)
k_trunk_scan(const TrGeom g, const TrTables t)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ uint64_t sh16[16];
    uint64_t c_nod = 0, c_blk = 0, c_sr = 0;             // carried over the rounds (the same in every lane)
    const uint32_t tid = threadIdx.x;
    for (uint32_t base = 0; base < g.nwin; base += 4096u) {
        const uint32_t w0 = base + tid * 4u;
        uint32_t nod[4], blk[4], ros[4];
        uint64_t prev[4], ent[4];
        uint32_t nsum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = w0 + j;
            const bool in = w < g.nwin;
            nod[j] = in ? t.ccnt[w] : 0u;
            blk[j] = in ? t.nblk[w] : 0u;
            ros[j] = in ? t.nros[w] : 0u;
            prev[j] = (in && w) ? t.exit[w - 1u] : kTrNone;
            ent[j] = in ? t.entry[w] : kTrNone;
            nsum += nod[j];
        }
        uint64_t tot;
        uint64_t P = c_nod + block_incl_scan64(nsum, sh16, tot) - nsum;      // nodes in front, unclipped
        c_nod += tot;
        uint64_t Pj[4], bsum = 0, srsum = 0;
        bool fits[4], seam[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = w0 + j;
            Pj[j] = P;
            fits[j] = P + nod[j] <= g.ncap;
            seam[j] = w < g.nwin && (w == 0u || prev[j] == kTrNone || prev[j] != ent[j] || P > g.ncap);
            P += nod[j];
            if (!fits[j]) {
                blk[j] = 0;
                ros[j] = 0;
            }
            bsum += blk[j];
            srsum += ((uint64_t)(seam[j] ? 1u : 0u) << 32) | ros[j];
        }
        uint64_t G = c_blk + block_incl_scan64(bsum, sh16, tot) - bsum;
        c_blk += tot;
        uint64_t SR = c_sr + block_incl_scan64(srsum, sh16, tot) - srsum;    // seams / runs in front of w0
        c_sr += tot;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = w0 + j;
            SR += (uint64_t)(seam[j] ? 1u : 0u) << 32;
            if (w < g.nwin) {
                if (!fits[j]) {
                    t.ccnt[w] = 0;
                    t.nblk[w] = 0;
                    t.nros[w] = 0;
                }
                t.nbase[w] = (uint32_t)(Pj[j] < g.ncap ? Pj[j] : g.ncap);
                t.gbase[w] = G;
                t.seampre[w] = (uint32_t)(SR >> 32);                          // seams at or in front of w
                t.rospre[w] = (uint32_t)SR;                                   // runs in front of w
            }
            G += blk[j];
            SR += ros[j];
        }
    }
    if (tid == 0) {
        t.nbase[g.nwin] = (uint32_t)(c_nod < g.ncap ? c_nod : g.ncap);
        t.gbase[g.nwin] = c_blk;
        t.seampre[g.nwin] = (uint32_t)(c_sr >> 32) + 1u;
        t.rospre[g.nwin] = (uint32_t)c_sr;
    }
}

// Hypothesis walks.  Every node of the core windows is tried as an RSI start.  One workgroup per group of
// `wpg` windows (persistent: the groups in turn); the group's stretch of the stream plus a margin behind it is
// staged in LDS with the trunk marks and, for short coded data sets (NX), the byte table of ordinary coded
// data sets (aec_trunk.h: TrStaged), so that a step of a walk touches no device memory.  The walks differ
// wildly in length (a node in the middle of an RSI is back on the trunk after a few coded data sets, one behind
// a true RSI start after the trunk's resynchronisation distance), so the loop is flat -- one coded data set per
// lane and round, a lane that finishes takes its next node -- and a round serves ONE kind of step, the kind
// most lanes wait for:
//   TABLE   the byte table has the length (an ordinary coded data set inside an RSI)
//   PARSE   the parse on the staged words (reference sample, zero-block runs, long coded data sets)
//   REMOTE  device memory: the lane's next node, or a walk that has left the stretch
// LDS: sw[bits / 32 + 8] u32 | bm[bits / 32] u32 | wnb[wpg + 1] u32 | wnp[wpg + 1] u32 | nx[bits] u8 (NX)
template <bool NX>
__global__ void __launch_bounds__(1024)
k_hyp_walk(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, uint32_t wpg, uint32_t margin,
           uint32_t ngroups)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t hyp_lds[];
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u;
    const TrGlobal mem{s, g, t};
    const bool pp = c.flags & F_PREPROCESS;
    uint32_t pool_cur = 0, pool_end = 0;                     // this wavefront's piece of the pool (same in every lane)
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t w0 = grp * wpg;
        const uint32_t w1 = w0 + wpg < g.ncore ? w0 + wpg : g.ncore;
        const uint64_t base = g.lo + (uint64_t)w0 * g.L;
        const uint32_t bits = (w1 - w0) * g.L + margin;
        const uint32_t nw = bits / 32u;
        uint32_t *sw = hyp_lds, *bm = sw + nw + 8u, *wnb = bm + nw, *wnp = wnb + wpg + 1u;
        uint8_t *nx = reinterpret_cast<uint8_t *>(wnp + wpg + 1u);
        __syncthreads();                                      // (the group before is done with the arrays)
        {
            const uint64_t bw0 = base >> 5, gw0 = (base - g.lo) >> 5, gwn = (uint64_t)g.nwin * (g.L / 32u);
            for (uint32_t i = tid; i < nw + 8u; i += nt) sw[i] = tr_word(s, bw0 + i);
            for (uint32_t i = tid; i < nw; i += nt) bm[i] = gw0 + i < gwn ? t.bitmap[gw0 + i] : 0u;
            // nodes of the group's windows: where they are stored, and how many lie in front inside the group
            if (tid <= w1 - w0) wnb[tid] = t.nbase[w0 + tid];
        }
        __syncthreads();
        if (tid <= w1 - w0) wnp[tid] = wnb[tid] - wnb[0];    // (nbase[w + 1] - nbase[w] = ccnt[w] after the scan)
        TrStaged st{mem, sw, bm, NX ? nx : nullptr, base, bits};
        if (NX) {
            for (uint32_t q = tid; q < bits; q += nt) {
                TrWin W;
                st.win(base + q, W);
                nx[q] = tr_fast_entry(s, c, base + q, W);
            }
        }
        __syncthreads();
        const uint32_t nnodes = wnp[w1 - w0];
        // ---- the lane's walks: nodes tid, tid + nt, ... of the group
        enum : uint32_t { NEW = 0, TABLE = 1, PARSE = 2, REMOTE = 3, IDLE = 4 };
        uint32_t m = tid, wi = 0, cls = NEW, e8 = 0;          // node number in the group, its window (relative)
        TrHyp h;
        auto finish = [&](uint32_t state) {                   // record of the node, on to the next one
            const uint64_t at = (uint64_t)wnb[wi] + (m - wnp[wi]);
            TrRec r{0u, 0u};
            uint32_t park = 0;
            const uint64_t d = h.pos - h.c;
            if (state == TR_DONE) {
                r.x = tr_rec_pack(d, h.k);
                r.y = r.x ? h.link : 0u;
            } else if (state == TR_LAND && d <= 0xFFFFFFFFull) {
                r.x = (uint32_t)d;
                r.y = h.link;
                park = kTrParked | (h.k << 16) | h.b;
            }
            t.rec[at] = r;
            t.park[at] = park;
            m += nt;
            cls = NEW;
        };
        auto classify = [&]() {                               // what the walk's next step needs
            if (!st.in(h.pos)) {
                cls = REMOTE;
            } else if (h.b == 0u && pp) {
                cls = PARSE;
            } else {
                const uint32_t r = (uint32_t)(h.pos - base);
                if (h.b != 0u && ((bm[r >> 5] >> (31u - (r & 31u))) & 1u)) {
                    finish(TR_LAND);
                } else if (h.steps >= g.budget) {
                    finish(TR_FAIL);
                } else {
                    e8 = NX ? nx[r] : 0u;
                    cls = e8 ? TABLE : PARSE;
                }
            }
        };
        auto after = [&](uint32_t r) {                        // r = result of tr_hyp_advance / tr_hyp_step
            if (r != TR_RUN) finish(r);
            else classify();
        };
        for (;;) {
            const uint32_t nT = (uint32_t)__popcll(__ballot(cls == TABLE));
            const uint32_t nP = (uint32_t)__popcll(__ballot(cls == PARSE));
            const uint32_t nR = (uint32_t)__popcll(__ballot(cls == NEW || cls == REMOTE));
            if (!(nT | nP | nR)) break;
            if (nR >= 16u || !(nT | nP)) {
                if (cls == NEW) {
                    if (m < nnodes) {
                        while (m >= wnp[wi + 1u]) wi++;
                        const uint32_t cp = t.cpos[(uint64_t)wnb[wi] + (m - wnp[wi])];
                        tr_hyp_start(c, h, base + (uint64_t)wi * g.L + cp);
                        classify();
                    } else {
                        cls = IDLE;
                    }
                } else if (cls == REMOTE) {
                    after(tr_hyp_step(s, c, g, mem, h));
                }
            } else if (nP >= 16u || !nT) {
                if (cls == PARSE) {
                    TrWin W;
                    st.win(h.pos, W);
                    uint32_t nb = 1;
                    const uint32_t len = tr_hyp_parse(s, c, h, W, nb);
                    if (!len) finish(TR_FAIL);
                    else after(tr_hyp_advance(c, g, st, h, len, nb));
                }
            } else if (cls == TABLE) {
                after(tr_hyp_advance(c, g, st, h, e8, 1u));
            }
            // RSI ends to record: the wavefront's lanes take entries of its piece of the pool together
            const bool want = cls != NEW && cls != IDLE && h.pend != 0u;
            const uint64_t wm = __ballot(want);
            if (wm) {
                const uint32_t n = (uint32_t)__popcll(wm);
                if (pool_cur + n > pool_end) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(t.pool_cnt, 256u);
                    pool_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    pool_end = pool_cur + 256u;
                }
                if (want) {
                    const uint32_t e = pool_cur + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
                    if (!tr_hyp_commit(g, t, h, e)) finish(TR_FAIL);
                }
                pool_cur += n;
            }
        }
    }
}

// The same walks for long coded data sets, whose walks run over more of the stream than LDS holds: every step
// reads device memory.  One wavefront per group of `wpg` windows (many nodes per lane, so that the flat loop
// evens out the walk lengths).  A step is ONE memory round trip -- the 256-bit window at the walk's position
// and the word of trunk marks leave together -- and everything that needs another one waits in the lane until
// 16 lanes of the wavefront do (or nothing else is left): a unary part that goes on behind the window, the
// mark at the end of an RSI, the record of a finished walk and the lane's next node.
__global__ void __launch_bounds__(64)
k_hyp_walk_mem(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, uint32_t wpg)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t w0 = blockIdx.x * wpg;
    if (w0 >= g.ncore) return;
    const uint32_t w1 = w0 + wpg < g.ncore ? w0 + wpg : g.ncore;
    uint32_t pool_cur = 0, pool_end = 0;
    enum : uint32_t { STEP = 0, LONG = 1, ENDRSI = 2, FIN = 3, IDLE = 4 };
    uint32_t w = w0, idx = lane, cls = FIN, fin = 0xFFu;      // (fin 0xFF: nothing to record yet, just take a node)
    TrHyp h;
    h.pend = 0;
    for (;;) {
        const uint32_t nS = (uint32_t)__popcll(__ballot(cls == STEP));
        const uint32_t nR = (uint32_t)__popcll(__ballot(cls == LONG || cls == ENDRSI || cls == FIN));
        if (!(nS | nR)) break;
        if (nR >= 16u || !nS) {
            if (cls == LONG) {                                // the whole parse, following the unary part through memory
                TrWin W;
                tr_win_load(s, h.pos, W);
                uint32_t nb = 1;
                const uint32_t len = tr_hyp_parse(s, c, h, W, nb);
                if (!len) {
                    cls = FIN;
                    fin = TR_FAIL;
                } else {
                    h.pos += len;
                    h.b += nb;
                    h.steps++;
                    cls = h.b == c.rsi ? ENDRSI : STEP;
                }
            }
            if (cls == ENDRSI) {
                const uint32_t r = tr_hyp_complete(c, g, h, tr_marked(g, t, h.pos));
                cls = r == TR_RUN ? STEP : FIN;
                fin = r;
            } else if (cls == FIN) {
                if (fin != 0xFFu) {
                    tr_hyp_finish(g, t, w, idx, h, fin);
                    idx += 64u;
                }
                cls = IDLE;
                while (w < w1) {
                    const uint32_t n = t.ccnt[w];
                    if (idx < n) {
                        tr_hyp_start(c, h, g.lo + (uint64_t)w * g.L + t.cpos[t.nbase[w] + idx]);
                        cls = STEP;
                        break;
                    }
                    idx -= n;
                    w++;
                }
            }
        } else if (cls == STEP) {
            TrWin W;
            tr_win_load(s, h.pos, W);
            const bool on = tr_marked(g, t, h.pos);
            if (h.b != 0u && on) {
                cls = FIN;
                fin = TR_LAND;
            } else if (h.steps >= g.budget) {
                cls = FIN;
                fin = TR_FAIL;
            } else {
                uint32_t nb = 1;
                bool more = false;
                const uint32_t len = tr_hyp_parse(s, c, h, W, nb, false, &more);
                if (!len) {
                    cls = more ? LONG : FIN;
                    fin = TR_FAIL;
                } else {
                    h.pos += len;
                    h.b += nb;
                    h.steps++;
                    if (h.b == c.rsi) cls = ENDRSI;
                }
            }
        }
        // RSI ends to record: the wavefront's lanes take entries of its piece of the pool together
        const bool want = cls == STEP && h.pend != 0u;
        const uint64_t wm = __ballot(want);
        if (wm) {
            const uint32_t n = (uint32_t)__popcll(wm);
            if (pool_cur + n > pool_end) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(t.pool_cnt, 256u);
                pool_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                pool_end = pool_cur + 256u;
            }
            if (want) {
                const uint32_t e = pool_cur + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
                if (!tr_hyp_commit(g, t, h, e)) {
                    cls = FIN;
                    fin = TR_FAIL;
                }
            }
            pool_cur += n;
        }
    }
}

// ---- coalescing hypothesis walks (aec_trunk.h section 2b) ------------------------------------------------
// Lists the kernels below hand on: walks to be finished from device memory, nodes left to the plain walk.
struct CoLists {
    uint2 *queue;              // {node record index, window}
    uint2 *plain;              // {window, index inside the window}
    uint32_t *counts;          // [0] queue entries, [1] plain entries (both may run beyond their capacity)
    uint32_t qcap, pcap;
    uint32_t over_plain;       // CO_OVER nodes go to the plain walk too (RSIs not much longer than the way to the trunk)
};

__device__ __forceinline__ void co_push(uint2 *list, uint32_t *count, uint32_t cap, uint2 v)
{
    const uint32_t i = atomicAdd(count, 1u);
    if (i < cap) list[i] = v;
}

// One workgroup per group of `wpg` windows (persistent: the groups in turn).  The group's stretch of the stream
// plus a margin is staged in LDS with the trunk marks, and a table of MARKS over its bit positions (one cell per
// 2^shift bits) that the walks fill as they go; a walk that meets a mark stops and is resolved through the owner
// after the workgroup's barrier.  Flat loop: one parse per lane and round, a lane that is through takes its next
// node.  LDS: sw[bits / 32 + 8] | bm[bits / 32] | cells[(bits >> shift) + 1] | wnb[wpg + 1] | wnp[wpg + 1] |
// recs[cap] (8 bytes each)
struct CoCells {
    uint32_t *v;
    __device__ __forceinline__ uint32_t claim(uint32_t i, uint32_t p) { return atomicCAS(&v[i], 0u, p); }
    __device__ __forceinline__ uint32_t peek(uint32_t i) const { return v[i]; }
};

// (walks set aside after `park_at` parses: a workgroup lives as long as its longest walk -- 64 parses where the
// average walk takes 12 -- and every wavefront held one of the long ones: 4 of 64 lanes busy per instruction.  The
// long walks are continued densely packed in the first wavefronts, the others go on to the barrier.)
constexpr uint32_t kCoPark = 256;
struct CoParked {
    uint32_t m, c_rel, pos_rel, bb;      // node, its position and the walk's from `base`, b | b_ros << 13
};

__global__ void __launch_bounds__(256)
k_hyp_walk_co(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, uint32_t wpg, uint32_t margin,
              uint32_t ngroups, uint32_t shift, uint32_t tmax, uint32_t cap, const CoLists ls, uint32_t park_at)
{
    if (t.skip_if && *t.skip_if) return;
    // ls.over_plain: walks that run over the end of their RSI before they land go to the plain walk as well
    extern __shared__ __attribute__((aligned(16))) uint32_t co_lds[];
    __shared__ CoParked pk[kCoPark];
    __shared__ uint32_t sh_npark;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const TrGlobal mem{s, g, t};
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t w0 = grp * wpg;
        const uint32_t w1 = w0 + wpg < g.ncore ? w0 + wpg : g.ncore;
        const uint64_t base = g.lo + (uint64_t)w0 * g.L;
        const uint32_t bits = wpg * g.L + margin, nw = bits / 32u, ncell = (bits >> shift) + 1u;
        uint32_t *sw = co_lds, *bm = sw + nw + 8u, *cellv = bm + nw, *wnb = cellv + ncell, *wnp = wnb + wpg + 1u;
        CoRec *recs = reinterpret_cast<CoRec *>(wnp + wpg + 1u + (ncell & 1u));          // (8-byte aligned)
        __syncthreads();                                      // (the group before is done with the arrays)
        {
            const uint64_t bw0 = base >> 5, gw0 = (base - g.lo) >> 5, gwn = (uint64_t)g.nwin * (g.L / 32u);
            for (uint32_t i = tid; i < nw + 8u; i += nt) sw[i] = tr_word(s, bw0 + i);
            for (uint32_t i = tid; i < nw; i += nt) bm[i] = gw0 + i < gwn ? t.bitmap[gw0 + i] : 0u;
            for (uint32_t i = tid; i < ncell; i += nt) cellv[i] = 0u;
            if (tid <= w1 - w0) wnb[tid] = t.nbase[w0 + tid];
            if (tid == 0) sh_npark = 0u;
        }
        __syncthreads();
        if (tid <= w1 - w0) wnp[tid] = wnb[tid] - wnb[0];
        __syncthreads();
        const uint32_t nnodes = wnp[w1 - w0], nown = nnodes < cap ? nnodes : cap;
        const TrStaged st{mem, sw, bm, nullptr, base, bits};
        CoCells cells{cellv};
        const uint64_t lim = base + bits - 1024u;
        // ---- the walks: nodes tid, tid + nt, ... of the group
        uint32_t m = tid, wi = 0;
        bool run = false;
        CoWalk h;
        auto finish = [&](uint32_t node, uint32_t r, uint32_t hit) {
            uint32_t tt = 0;
            if (r == CO_LAND || r == CO_QUEUE || r == CO_OVER) {
                if (h.pos - h.c > 0xFFFFFFFFull) r = CO_PLAIN;
                tt = (uint32_t)(h.pos - h.c);
            } else if (r == CO_LINK) {
                tt = hit;
            }
            recs[node] = CoRec{co_pack(r, h.b, h.b_ros), tt};
        };
        for (;;) {
            if (!run && m < nown) {
                while (m >= wnp[wi + 1u]) wi++;
                const uint32_t cp = t.cpos[(uint64_t)wnb[wi] + (m - wnp[wi])];
                tr_co_start(c, h, base + (uint64_t)wi * g.L + cp);
                run = true;
            }
            if (!__any(run)) break;
            if (run) {
                uint32_t hit = 0;
                const uint32_t r = tr_co_step(s, c, st, cells, h, m, base, lim, shift, tmax, hit);
                if (r != CO_RUN) {
                    finish(m, r, hit);
                    m += nt;
                    run = false;
                } else if (h.steps == park_at) {
                    const uint32_t slot = atomicAdd(&sh_npark, 1u);
                    if (slot < kCoPark) {
                        pk[slot] = CoParked{m, (uint32_t)(h.c - base), (uint32_t)(h.pos - base), h.b | (h.b_ros << 13)};
                        m += nt;
                        run = false;
                    }
                }
            }
        }
        __syncthreads();
        {
            const uint32_t np = sh_npark < kCoPark ? sh_npark : kCoPark;
            uint32_t j = tid, node = 0;
            run = false;
            for (;;) {
                if (!run && j < np) {
                    const CoParked q = pk[j];
                    node = q.m;
                    h.c = base + q.c_rel;
                    h.pos = base + q.pos_rel;
                    h.b = q.bb & 0x1FFFu;
                    h.b_ros = q.bb >> 13;
                    h.steps = park_at;
                    j += nt;
                    run = true;
                }
                if (!__any(run)) break;
                if (run) {
                    uint32_t hit = 0;
                    const uint32_t r = tr_co_step(s, c, st, cells, h, node, base, lim, shift, tmax, hit);
                    if (r != CO_RUN) {
                        finish(node, r, hit);
                        run = false;
                    }
                }
            }
        }
        __syncthreads();
        // ---- every node's links followed inside the group; what it comes to goes into its record
        wi = 0;
        for (m = tid; m < nnodes; m += nt) {
            while (m >= wnp[wi + 1u]) wi++;
            const uint32_t idx = m - wnp[wi];
            const uint64_t at = (uint64_t)wnb[wi] + idx;
            uint64_t z = 0;
            uint32_t root = 0, b = 0, kind = CO_PLAIN;
            auto node_pos = [&](uint32_t q) -> uint64_t {
                uint32_t v = 0;
                while (q >= wnp[v + 1u]) v++;
                return base + (uint64_t)v * g.L + t.cpos[(uint64_t)wnb[v] + (q - wnp[v])];
            };
            if (m < nown) kind = tr_co_resolve(c, [&](uint32_t q) { return recs[q]; }, node_pos, m, z, root, b);
            TrRec out{0u, 0u};
            uint32_t park = 0;
            const uint64_t mine = base + (uint64_t)wi * g.L + t.cpos[at];
            if (kind == CO_LAND && z - mine <= 0xFFFFFFFFull) {
                out.x = (uint32_t)(z - mine);
                park = kTrParked | b;                          // (a parked hypothesis: k = 0, b blocks done)
            } else if (kind == CO_QUEUE) {
                out.x = recs[m].t;
                park = recs[m].k;
                co_push(ls.queue, ls.counts, ls.qcap, make_uint2((uint32_t)at, w0 + wi));
            } else if (kind == CO_GOON && z > mine && z - mine <= 0xFFFFFFFFull) {
                out.x = (uint32_t)(z - mine);                  // (handed on like a root, from where the other walk ran over)
                park = co_pack(CO_QUEUE, b, kCoNoRos);
                co_push(ls.queue, ls.counts, ls.qcap, make_uint2((uint32_t)at, w0 + wi));
                kind = CO_QUEUE;
            } else if (kind == CO_DEFER) {
                // the walk it waits for: its record, the count that one was handed on with, and where its node lies
                // from here (the distance its landing will be given by is from ITS node)
                const int64_t delta = (int64_t)node_pos(root) - (int64_t)mine;
                uint32_t rv = 0;
                while (root >= wnp[rv + 1u]) rv++;
                out.x = wnb[rv] + (root - wnp[rv]);
                out.y = co_bend(recs[root].k) | ((uint32_t)(delta + 0x40000) << 13);
                park = co_pack(CO_DEFER, b, kCoNoRos);
                if (delta < -0x40000 || delta >= 0x40000) kind = CO_PLAIN;
            }
            if (kind != CO_LAND && kind != CO_QUEUE && kind != CO_DEFER) {
                out = TrRec{0u, 0u};
                park = 0;
                if (kind == CO_PLAIN || (kind == CO_OVER && ls.over_plain))
                    co_push(ls.plain, ls.counts + 1, ls.pcap, make_uint2(w0 + wi, idx));
            } else if (kind == CO_LAND && !park) {
                co_push(ls.plain, ls.counts + 1, ls.pcap, make_uint2(w0 + wi, idx));
            }
            t.rec[at] = out;
            t.park[at] = park;
        }
    }
}

// the walks that were handed on: one lane each, from device memory until they stand on the trunk.  (A wavefront per
// walk, aec_coop.h, was measured here and lost: 11.3 ms against 3.8 per span of 2 Gbit -- a wavefront's parse is
// mostly scalar work, the CU's one scalar unit serves all its wavefronts, and there are sixty thousand of these
// walks; the lanes' time is the longest walk's, ~1500 coded data sets at 2.5 us.)
constexpr uint32_t kCoopWin = 1024;                  // words of stream (and of trunk marks) a wavefront stages (k_seg_starts)
__global__ void __launch_bounds__(64)
k_hyp_walk_rest(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, const CoLists ls)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t n = ls.counts[0] < ls.qcap ? ls.counts[0] : ls.qcap;
    const TrGlobal mem{s, g, t};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 e = ls.queue[i];
        const uint64_t at = e.x;
        const uint64_t node = g.lo + (uint64_t)e.y * g.L + t.cpos[at];
        const uint32_t pk = t.park[at], bh = co_bend(pk);
        uint32_t dist = 0;
        const uint32_t res = tr_co_rest(s, c, g, mem, node, node + t.rec[at].x, bh, co_bros(pk), dist);
        t.rec[at] = TrRec{dist, bh};                           // (y: the count it was handed on with -- k_hyp_defer)
        t.park[at] = res;
        const uint32_t rk = co_kind(res);
        if (rk == CO_PLAIN || (rk == CO_OVER && (ls.over_plain || co_bend(res))))
            co_push(ls.plain, ls.counts + 1, ls.pcap, make_uint2(e.y, (uint32_t)(at - t.nbase[e.y])));
    }
}

// one lane per node: the nodes that waited for a handed-on walk take its landing
__global__ void __launch_bounds__(256)
k_hyp_defer(const Cfg c, const TrGeom g, const TrTables t, const CoLists ls)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t w = blockIdx.x;
    if (w >= g.ncore) return;
    const uint32_t n = t.ccnt[w];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t at = t.nbase[w] + i;
        const uint32_t pk = t.park[at];
        if ((pk & kTrParked) || !(pk & kCoTag) || co_kind(pk) != CO_DEFER) continue;
        const TrRec r = t.rec[at];
        const uint32_t root_pk = t.park[r.x];
        uint32_t kind;
        const uint32_t b = co_defer(c, co_bend(pk), r.y & 0x1FFFu, root_pk, kind);
        const int64_t dist = (int64_t)(r.y >> 13) - 0x40000 + (int64_t)t.rec[r.x].x;
        if (b && dist > 0 && dist <= 0xFFFFFFFFll) {
            t.rec[at] = TrRec{(uint32_t)dist, 0u};
            t.park[at] = kTrParked | b;
        } else {
            t.rec[at] = TrRec{0u, 0u};
            t.park[at] = 0;
            if (kind == CO_LAND) kind = CO_PLAIN;
            if (kind == CO_PLAIN || (kind == CO_OVER && ls.over_plain)) co_push(ls.plain, ls.counts + 1, ls.pcap, make_uint2(w, i));
        }
    }
}

// the nodes left to the plain walk (tr_hyp_step: any number of RSIs, the pool of RSI ends), walk and jump.  A
// WAVEFRONT per node (aec_coop.h): these are few -- the true RSI starts behind which the trunk has not found the
// true chain again within a whole RSI, and what shares their way -- and their walks are thousands of coded data
// sets long (a lane each took as long as the serial walker needs for them, 2.5 ms per RSI).
__global__ void __launch_bounds__(64)
k_hyp_walk_list(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, const CoLists ls)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ __attribute__((aligned(16))) uint32_t lds_w[kCoopWin], lds_m[kCoopWin];
    const uint32_t n = ls.counts[1] < ls.pcap ? ls.counts[1] : ls.pcap;
    const TrGlobal mem{s, g, t};
    CoopCds<kCoopWin> cw;
    cw.init(s, c, lds_w);
    cw.with_marks(lds_m, g, t);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint2 e = ls.plain[i];
        TrHyp h;
        tr_hyp_start(c, h, g.lo + (uint64_t)e.x * g.L + t.cpos[t.nbase[e.x] + e.y]);
        uint32_t st = TR_RUN;
        if (!cw.usable()) {
            if (threadIdx.x == 0) {
                while ((st = tr_hyp_step(s, c, g, mem, h)) == TR_RUN)
                    if (h.pend && !tr_hyp_commit(g, t, h, atomicAdd(t.pool_cnt, 1u))) {
                        st = TR_FAIL;
                        break;
                    }
                tr_hyp_finish(g, t, e.x, e.y, h, st);
                tr_hyp_land(c, g, t, e.x, e.y);
            }
            continue;
        }
        // (tr_hyp_step, the whole wavefront in step; every value below is the same in all lanes)
        while (st == TR_RUN) {
            const bool first = h.b == 0u;
            cw.prepare(h.pos);
            if (!first && cw.marked(h.pos)) {
                st = TR_LAND;
                break;
            }
            if (h.steps >= g.budget) {
                st = TR_FAIL;
                break;
            }
            const uint32_t ref = (first && (c.flags & F_PREPROCESS)) ? 1u : 0u;
            uint32_t nz;
            const uint32_t len = cw.cds(c, h.pos, ref, nz);
            const uint32_t nb = len ? tr_blocks(c, nz, h.b) : 0u;
            if (!nb || nb > c.rsi - h.b) {
                st = TR_FAIL;
                break;
            }
            h.pos += len;
            h.b += nb;
            h.steps++;
            if (h.b == c.rsi) {
                st = tr_hyp_complete(c, g, h, tr_marked(g, t, h.pos));
                if (st == TR_RUN && h.pend) {
                    uint32_t slot = 0;
                    if (threadIdx.x == 0) slot = atomicAdd(t.pool_cnt, 1u);
                    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
                    if (slot >= g.pcap) {
                        st = TR_FAIL;
                    } else {
                        if (threadIdx.x == 0) t.pool[slot] = TrPoolEntry{h.pend, h.link};
                        h.link = slot + 1u;
                        h.pend = 0;
                    }
                }
            }
        }
        if (threadIdx.x == 0) {
            tr_hyp_finish(g, t, e.x, e.y, h, st);
            __threadfence();
            tr_hyp_land(c, g, t, e.x, e.y);
        }
    }
}

// one lane per node: the jump of every parked hypothesis
__global__ void __launch_bounds__(256)
k_hyp_land(const Cfg c, const TrGeom g, const TrTables t)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t w = blockIdx.x;
    if (w >= g.ncore) return;
    const uint32_t n = t.ccnt[w];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) tr_hyp_land(c, g, t, w, i);
}

// ---- wide walker: every node of a chunk's first windows chases the records through the chunk.  "First windows":
// as many as hold an RSI, so that the true chain -- whose RSI starts lie an RSI apart -- has a start among them in
// every chunk, and the walk needs a few lookups per CHUNK, not one per RSI (RSIs of a megabit: one dependent read
// each, 3.4 us, 24 ms for the 7000 RSIs of a 7-Gbit span before).
__global__ void __launch_bounds__(256)
k_twide(const TwTables t, uint32_t nwin, uint64_t end_bit, uint4 *__restrict__ wide)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t chunk = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t first = chunk * t.wpc;
    if (first >= nwin || i >= t.wcap) return;
    const uint32_t lastf = first + t.wfirst < nwin ? first + t.wfirst : nwin;
    const uint32_t at = t.nbase[first] + i;
    if (at >= t.nbase[lastf]) return;
    uint32_t w = first, hi = lastf;                     // the node's window: nbase[w] <= at < nbase[w + 1]
    while (hi - w > 1u) {
        const uint32_t mid = w + (hi - w) / 2u;
        if (t.nbase[mid] <= at) w = mid;
        else hi = mid;
    }
    const uint32_t last = first + t.wpc < nwin ? first + t.wpc : nwin;      // one past the chunk's windows
    const uint64_t stop = t.lo + (uint64_t)last * t.core;
    uint64_t pos = t.lo + (uint64_t)w * t.core + t.cpos[at];
    uint32_t cnt = 0, ok = 1;
    while (pos < stop && pos < end_bit) {
        TrRec rec;
        uint32_t wv, ix;
        if (!tw_lookup(t, pos, rec, wv, ix) || !rec.x) {
            ok = 0;
            break;
        }
        pos += tr_rec_bits(rec.x);
        cnt += tr_rec_k(rec.x);
    }
    // (a chain that ends at the end of the input inside the last chunk is resolved as far as it goes:
    // the walker takes over from its exit)
    wide[(uint64_t)chunk * t.wcap + i] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), cnt, ok && cnt ? 1u : 0u);
}

// RSI starts of one record at node p (RSI number r onwards): the record's own start, and for a record of
// several RSIs the starts inside it from the pool of RSI ends
__device__ __forceinline__ void write_record_starts(const Cfg &c, const TrTables &tt, uint64_t p, uint64_t r, TrRec rec,
                                                    uint64_t *__restrict__ rsi_off)
{
    const uint32_t k = tr_rec_k(rec.x);
    rsi_off[r] = tr_rsi_start(c, p);
    if (k > 1u) (void)tr_rec_ends(tt, p, k, rec.y, [&](uint32_t j, uint64_t q) { rsi_off[r + 1u + j] = tr_rsi_start(c, q); });
}

// the true chain through the chunks the walker skipped: one lane per chunk follows the records and writes
// the RSI starts
__global__ void __launch_bounds__(64)
k_trewalk(const Cfg c, const TwTables t, const TrTables tt, uint32_t nwin, uint32_t nchunks, uint64_t end_bit,
         const ChunkEntry *__restrict__ entry, uint64_t *__restrict__ rsi_off)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= nchunks || !entry[chunk].valid) return;
    const uint32_t first = chunk * t.wpc;
    const uint32_t last = first + t.wpc < nwin ? first + t.wpc : nwin;
    const uint64_t stop = t.lo + (uint64_t)last * t.core;
    uint64_t pos = entry[chunk].pos, r = entry[chunk].r;
    while (pos < stop && pos < end_bit) {
        TrRec rec;
        uint32_t wv, ix;
        if (!tw_lookup(t, pos, rec, wv, ix) || !rec.x) break;      // (cannot happen: k_twide went through)
        write_record_starts(c, tt, pos, r, rec, rsi_off);
        pos += tr_rec_bits(rec.x);
        r += tr_rec_k(rec.x);
    }
}

// RSI starts inside the records of several RSIs the walker took outside the wide hops
__global__ void k_texpand(const Cfg c, const TwTables t, const TrTables tt, const IdxCarry *__restrict__ carry,
                         const IdxHop *__restrict__ hops, uint64_t *__restrict__ rsi_off)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= carry->n_hops) return;
    const IdxHop h = hops[i];
    TrRec rec;
    uint32_t wv, ix;
    if (!tw_lookup(t, h.pos, rec, wv, ix) || tr_rec_k(rec.x) != h.cnt) return;      // (cannot happen)
    write_record_starts(c, tt, h.pos, h.r, rec, rsi_off);
}

// Segment starts of the RSIs the walk found in this span (aec_trunk.h: tr_seg_walk, tr_jump_to), for decoders
// that take a lane per segment (aec_dec.hip: launch_decode_bare): seg_bits[r * segs_per_rsi + j] = start bit of
// segment j of RSI r; entries it cannot give stay as the caller initialised them (~0).  One wavefront per RSI:
// lane 0 walks from the RSI start until it stands on the trunk, then every lane takes one of the remaining
// segments -- a search in the block numbering of the trunk each.  The walk is the wavefront's (aec_coop.h: a few
// hundred coded data sets, some a few thousand, and the longest one sets the kernel's time).
__global__ void __launch_bounds__(64)
k_seg_starts(const Cfg c, const TrStream s, const TrGeom g, const TrTables t, const uint64_t *__restrict__ rsi_off,
             const IdxCarry *__restrict__ carry, const DecResult *__restrict__ res, uint64_t *__restrict__ seg_bits,
             uint64_t cap_rsi)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ __attribute__((aligned(16))) uint32_t lds_w[kCoopWin], lds_m[kCoopWin];
    const TrGlobal mem{s, g, t};
    CoopCds<kCoopWin> cw;
    cw.init(s, c, lds_w);
    cw.with_marks(lds_m, g, t);
    const uint32_t lane = threadIdx.x, S = c.segs_per_rsi;
    const uint64_t r0 = carry->r_prev;
    const bool on = carry->active != 0u;                    // (the walk goes on in the next span: all RSIs whole)
    const uint64_t whole = on ? carry->r : res->n_rsi;
    const uint32_t tail = on ? 0u : (uint32_t)res->tail_blocks;
    const uint64_t r1 = whole + (tail ? 1u : 0u);
    for (uint64_t r = r0 + blockIdx.x; r < r1 && r < cap_rsi; r += gridDim.x) {
        const uint32_t nblocks = r < whole ? c.rsi : tail;
        uint64_t *out = seg_bits + r * S;
        uint64_t pos = rsi_off[r];
        uint32_t b = 0;
        bool ok = true;
        if (!cw.usable()) {
            uint32_t st = 0;
            if (lane == 0)
                st = tr_seg_walk(s, c, mem, pos, b, nblocks, ~0ull, [&](uint32_t j, uint64_t q) { out[j] = q; });
            ok = __builtin_amdgcn_readfirstlane((int)st) == (int)TR_SEG_ON_TRUNK;
            pos = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pos >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pos);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        } else {
            // (aec_trunk.h tr_seg_walk, the whole wavefront in step)
            if (lane == 0) out[0] = pos;
            while (b < nblocks) {
                cw.prepare(pos);
                if (b != 0u && cw.marked(pos)) break;
                const uint32_t ref = (b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
                uint32_t nz;
                const uint32_t len = cw.cds(c, pos, ref, nz);
                const uint32_t nb = len ? tr_blocks(c, nz, b) : 0u;
                if (!nb || nb > nblocks - b || (b % 64u) + nb > 64u) {
                    ok = false;
                    break;
                }
                pos += len;
                b += nb;
                if ((b % 64u) == 0u && b < nblocks && lane == 0) out[b / 64u] = pos;
            }
        }
        if (!ok) {
            if (lane == 0) out[0] = ~0ull;                  // (no first segment = the RSI is not taken by segments)
            continue;
        }
        // segments b / 64 + 1 ... : one jump each
        const uint32_t nseg = (nblocks + 63u) / 64u;
        bool miss = false;
        for (uint32_t j = b / 64u + 1u + lane; j < nseg; j += 64u) {
            const uint64_t e = tr_jump_to(c, g, t, pos, b, j * 64u);
            out[j] = e == kTrNone ? ~0ull : e;
            miss = miss || e == kTrNone;
        }
        // A jump that does not come out (a seam of the trunk on the way: small inputs, whose regions are short): the
        // walk goes on to the end of the RSI instead and leaves every segment start it passes -- a millisecond, where
        // the decoder would take the whole RSI with one lane, twenty.
        if (__any(miss) && cw.usable()) {
            bool ok2 = true;
            while (b < nblocks) {
                uint32_t nz;
                const uint32_t len = cw.cds(c, pos, 0u, nz);
                const uint32_t nb = len ? tr_blocks(c, nz, b) : 0u;
                if (!nb || nb > nblocks - b || (b % 64u) + nb > 64u) {
                    ok2 = false;
                    break;
                }
                pos += len;
                b += nb;
                if ((b % 64u) == 0u && b < nblocks && lane == 0) out[b / 64u] = pos;
            }
            if (!ok2 && lane == 0) out[0] = ~0ull;
        }
    }
}

// ======== sparse candidates per window (low-entropy streams with short RSIs) ==========================
// ---- sparse speculation (aec_spec2.h) ----------------------------------------------------------------
// One workgroup of 1024 lanes per window = [lead-in | core | look-ahead].  LDS:
//   win[nw + 2] u32 | marks[nw] u32 | rank[nw + 2] u16 | sel[nw + 2] u16 | mpre[nw + 2] u16 |
//   cpos[cap] u16 | cnxt[cap] u16 | chop4[cap] u16 | chop16[cap] u16 | ua[cap] u16
// Output for the candidates inside the core: the window's part of the global bitmap / prefix table
// and its records (SparseTables above).
constexpr uint32_t kS2BridgeCap = 1u << 18;  // hypotheses per span k_bridge takes on (more: left to the walker)
constexpr uint32_t kS2FlatWindows = 48;     // spans of at most this many windows: no chunk-level walkers
constexpr uint32_t kS2BridgeAfter = 16;      // RSIs the walker had to walk itself before k_bridge steps in

struct Spec2Geom {
    uint32_t lead, core, look, stride, burn, cap_lds, cap_core, fast, refill, uncrun;
    uint32_t v4;          // k_spec4 (a table of every position's coded data set) instead of k_spec2
    const uint32_t *skip_if = nullptr;   // != 0 there: a scheme in front has delivered the stream; the kernel returns at once
};

__global__ void __launch_bounds__(1024)
k_spec2(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, uint64_t tab_lo,
        uint64_t start_bit, const Spec2Geom g, uint32_t *__restrict__ gbitmap, uint16_t *__restrict__ gpre,
        uint2 *__restrict__ grec, uint16_t *__restrict__ gcpos, uint32_t *__restrict__ gccnt,
        unsigned long long *__restrict__ prof, const uint64_t *__restrict__ starts = nullptr, uint32_t nstarts = 0,
        uint32_t *__restrict__ blist = nullptr, uint32_t *__restrict__ blist_cnt = nullptr)
{
    if (g.skip_if && *g.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t spec_lds[];
    __shared__ uint32_t sh_total, sh_next[2], sh_part[16];
    // (diagnostic builds of the host pass `prof`: shader-clock stamps at the phase boundaries)
    auto stamp = [&](int k) {
        if (prof && threadIdx.x == 0) prof[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const uint32_t W = g.lead + g.core + g.look, nw = W / 32u, cap = g.cap_lds;
    uint32_t *win = spec_lds;
    uint32_t *marks = win + nw + 4;                   // (spec_cds_fast reads four words from any position)
    uint16_t *rank = reinterpret_cast<uint16_t *>(marks + nw);
    uint16_t *sel = rank + nw + 2;
    uint16_t *mpre = sel + nw + 2;
    uint16_t *cpos = mpre + nw + 2;
    uint16_t *cnxt = cpos + cap;
    uint16_t *chop4 = cnxt + cap;
    uint16_t *chop16 = chop4 + cap;
    uint16_t *csucc = chop16 + cap;
    uint16_t *ua = csucc + cap;
    uint16_t *ub = ua + cap;
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
    for (uint32_t i = tid; i < nw + 4; i += nt) {
        const uint64_t idx = w0 + i;
        win[i] = idx < nwords ? bswap32(words[idx]) : 0u;
        if (i < nw) marks[i] = 0u;
    }
    __syncthreads();
    // prefix counts of 1-bits (and, further down, of marks) per word: every wavefront scans its slice of the
    // words (64 per round), then adds what the slices in front of it hold  (contains two barriers)
    auto prefix16 = [&](const uint32_t *src, uint16_t *dst) {
        const uint32_t nwv = nt >> 6, wv = tid >> 6, ln = tid & 63u;
        const uint32_t per = ((nw + nwv - 1u) / nwv + 63u) & ~63u;          // words per wavefront
        const uint32_t lo = wv * per, hi = lo + per < nw ? lo + per : nw;
        uint32_t carry = 0;
        for (uint32_t base = lo; base < hi; base += 64) {
            const uint32_t i = base + ln;
            const uint32_t pc = i < hi ? (uint32_t)__popc(src[i]) : 0u;
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < hi) dst[i + 1] = (uint16_t)(carry + incl);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (ln == 0) sh_part[wv] = carry;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t v = 0; v < nwv; v++) {
            const uint32_t t = sh_part[v];
            before += v < wv ? t : 0u;
            all += t;
        }
        for (uint32_t i = lo + ln; i < hi; i += 64) dst[i + 1] = (uint16_t)(dst[i + 1] + before);
        if (tid == 0) {
            dst[0] = 0;
            sh_total = all;
        }
        __syncthreads();
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
    // a coded data set without / with (`ref`) a reference sample: the straight-line parse, the full one where a
    // unary part does not end inside its 64-bit peek (rare at the coded rates this kernel is chosen for)
    const bool use_fast = g.fast != 0;
    // (a walk that fails this close to the end of the window may have run out of it -- see kS2Limited)
    auto near_end = [&](uint32_t pos) { return pos + 4096u > s.limit; };
    auto cds1 = [&](uint32_t q, uint32_t &run) -> uint32_t {
        if (!use_fast) return spec_cds(s, c, q, 1u, run);
        uint32_t len = spec_cds_fast<1>(win, s.limit, c, q, run);
        if (len == kSpecUnresolved) len = spec_cds(s, c, q, 1u, run);
        return len;
    };
    auto cds0 = [&](uint32_t q, uint32_t &run) -> uint32_t {
        if (!use_fast) return spec_cds(s, c, q, 0u, run);
        uint32_t len = spec_cds_fast<0>(win, s.limit, c, q, run);
        if (len == kSpecUnresolved) len = spec_cds(s, c, q, 0u, run);
        return len;
    };

    // ---- 1. sync chains: burn in, then mark until a marked boundary is met
    if (tid == 0 && start_bit >= wstart && start_bit - wstart < s.limit) {
        const uint32_t q = (uint32_t)(start_bit - wstart);
        atomicOr(&marks[q >> 5], 1u << (31u - (q & 31u)));                  // the one boundary that is known
    }
    if (starts && tid == 0) {
        // a batch of independent streams in one buffer (byte offsets `starts`): every stream begins with an RSI
        uint32_t lo = 0, hi = nstarts;
        while (lo < hi) {                                                   // first stream at or behind the window start
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (starts[mid] * 8u < wstart) lo = mid + 1u;
            else hi = mid;
        }
        for (; lo < nstarts && starts[lo] * 8u - wstart < s.limit; lo++) {
            const uint32_t q = (uint32_t)(starts[lo] * 8u - wstart);
            atomicOr(&marks[q >> 5], 1u << (31u - (q & 31u)));
        }
    }
    for (uint32_t q0 = tid * g.stride; q0 < s.limit; q0 += nt * g.stride) {
        uint32_t q = q0;
        bool ok = true;
        uint32_t run;
        for (uint32_t k = 0; k < g.burn && ok; k++) {
            const uint32_t len = q < s.limit ? cds0(q, run) : 0u;
            ok = len != 0;
            q += len;
        }
        while (ok && q < s.limit) {
            const uint32_t bit = 1u << (31u - (q & 31u));
            if (atomicOr(&marks[q >> 5], bit) & bit) break;
            const uint32_t len = cds0(q, run);
            ok = len != 0;
            q += len;
        }
    }
    // ---- 1b. runs of uncompressed coded data sets (an RSI of incompressible data).  Their headers -- all ones -- lie a
    // fixed distance apart, so they are found without a chain that leads there.  A window that begins inside such an
    // RSI needs that: its own chains start in noise, 24 coded data sets of burn-in are more than its lead-in holds,
    // and the RSI start BEHIND the run would stay unmarked (the walker then takes that RSI coded data set by coded
    // data set, and a chunk with one such RSI loses its chunk-level lookup).  Eight headers in a row start a marking
    // chain; a false alarm on other data marks a few garbage boundaries until it meets the marked chain, which costs
    // candidates and nothing else.
    {
        const uint32_t il = c.id_len, step = il + c.bs * c.bps;
        // (bit-parallel: for the 32 positions of a word, "id_len ones start here" as a mask; the same for the positions
        // one, two and three coded data sets further on; their AND leaves next to nothing on other data)
        // (all words of the four places are read before any is looked at: one LDS round trip per word of the window)
        auto ones_of = [&](uint32_t a, uint32_t b, uint32_t d, uint32_t sh) -> uint32_t {   // 32 positions from bit sh of a:b:d
            const uint64_t hi = ((uint64_t)a << 32) | b;
            const uint64_t y = sh ? (hi << sh) | ((uint64_t)d >> (32u - sh)) : hi;
            uint64_t m = y;
            for (uint32_t k = 1; k < il; k++) m &= y << k;
            return (uint32_t)(m >> 32);
        };
        const uint32_t w1 = step >> 5, s1 = step & 31u, w2 = (2u * step) >> 5, s2 = (2u * step) & 31u,
                       w3 = (3u * step) >> 5, s3 = (3u * step) & 31u;
        for (uint32_t i = tid; i < nw && g.uncrun; i += nt) {
            if (i * 32u + 32u + 7u * step + il > s.limit) break;
            const uint32_t a0 = win[i], b0 = win[i + 1u];
            const uint32_t a1 = win[i + w1], b1 = win[i + w1 + 1u], d1 = win[i + w1 + 2u];
            const uint32_t a2 = win[i + w2], b2 = win[i + w2 + 1u], d2 = win[i + w2 + 2u];
            const uint32_t a3 = win[i + w3], b3 = win[i + w3 + 1u], d3 = win[i + w3 + 2u];
            uint32_t hits = ones_of(a0, b0, 0u, 0u) & ones_of(a1, b1, d1, s1) & ones_of(a2, b2, d2, s2) & ones_of(a3, b3, d3, s3);
            // (four in a row still happen on compressible data -- 8-bit samples, 3-bit headers: a hundred per window, each
            // a chain of garbage boundaries: eight in a row it is, the second four only where the first four hold)
            for (uint32_t k = 4; k < 8u && hits; k++) {
                const uint32_t wk = i + ((k * step) >> 5);
                hits &= ones_of(win[wk], win[wk + 1u], win[wk + 2u], (k * step) & 31u);
            }
            while (hits) {
                const uint32_t j = (uint32_t)__builtin_clz(hits);
                hits &= ~(0x80000000u >> j);
                uint32_t q = i * 32u + j;
                bool ok = true;
                uint32_t run;
                while (ok && q < s.limit) {
                    const uint32_t bit = 1u << (31u - (q & 31u));
                    if (atomicOr(&marks[q >> 5], bit) & bit) break;
                    const uint32_t len = cds0(q, run);
                    ok = len != 0;
                    q += len;
                }
            }
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
        uint32_t run = 0;
        const uint32_t len = q < s.limit ? cds0(q, run) : 0u;
        cnxt[i] = len ? (uint16_t)(len | (run ? kNxtZero : kNxtBlock)) : (uint16_t)0;
    }
    __syncthreads();
    stamp(3);
    S2Win w{s, marks, mpre, cnxt, csucc, chop4, chop16, cpos, ncand};
    for (uint32_t i = tid; i < ncand; i += nt) csucc[i] = s2_succ(w, i);
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) chop4[i] = s2_hop4(w, c, i);
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) chop16[i] = s2_hop16(w, i);
    __syncthreads();
    stamp(4);
    // ---- 3. the RSI hypothesis at every candidate of the core
    const uint32_t i0 = mpre[c0 >> 5], i1 = mpre[c1 >> 5 < nw ? c1 >> 5 : nw];   // (c0, c1 are multiples of 32)
    const uint32_t ncore = i1 - i0;
    // (Flattening this loop -- one step per lane and iteration, a finished lane taking its next candidate
    // at once -- was measured and is SLOWER here, 521k against 430k cycles per window: the iterations of
    // the merged loop pay for every path, the on-demand parse included.)
    // s2_unit (aec_spec2.h) in two homogeneous passes, because a wavefront's loop costs what its slowest lane
    // costs (a walk off the marked chain takes 40 parses where the average takes 10) and a loop body that holds
    // both kinds of step pays for both: A. the parses -- first CDS with the reference sample, then on demand
    // until the walk stands on a marked boundary -- one per lane and round, a lane that is through fetching the
    // next candidate at once; B. the table steps from there, in the same manner.  Between the passes a candidate
    // keeps (distance, blocks) in ua / ub; 0 in ua = unresolved.
    if (tid == 0) sh_next[0] = sh_next[1] = i0;
    // A0. the first coded data set of every hypothesis (with the reference sample): one per lane, all alike
    {
        const uint32_t ref_first = (c.flags & F_PREPROCESS) ? 1u : 0u, bend = c.rsi;
        for (uint32_t i = i0 + tid; i < i1; i += nt) {
            const uint32_t p0 = cpos[i];
            // (ub = kS2Limited with ua = 0: the walk ran out of the window -- an RSI longer than the look-ahead -- as
            // opposed to a hypothesis the data refutes: k_bridge parses those from the stream)
            uint32_t delta = 0, bb = near_end(p0) ? kS2Limited : 0u;     // (a first coded data set that does not fit)
            if (p0 < s.limit) {
                uint32_t run;
                const uint32_t len = ref_first ? cds1(p0, run) : cds0(p0, run);
                bool fail = len == 0u;
                uint32_t n = 1;
                if (!fail && run) {
                    n = spec_run_blocks(c, len - c.id_len - 1u - (ref_first ? c.bps : 0u), 0u);
                    fail = !n || n > bend;
                    if (fail) bb = 0;
                }
                if (!fail) {
                    const uint32_t pos = p0 + len;
                    bb = pos >= s.limit ? kS2Limited : 0u;
                    if (n >= bend || (pos < s.limit && s2_marked(marks, pos))) {
                        delta = len;
                        bb = n;
                    } else if (pos < s.limit) {
                        delta = len;
                        bb = n | 0x8000u;                       // still off the marked chain: pass A1
                    }
                }
            }
            ua[i] = (uint16_t)delta;
            ub[i] = (uint16_t)bb;
        }
    }
    __syncthreads();
    // A1. catch-up: on-demand parses (no reference sample) until the walk stands on a marked boundary
    {
        const uint32_t bend = c.rsi;
        uint32_t i = 0, p0 = 0, pos = 0, b = 0;
        bool have = false, out = false;
        while (true) {
            // (taking the next candidate costs three dependent LDS round trips: idle lanes wait until kS2Refill of
            // them can do it together -- or nothing else is left to do)
            const uint64_t idle = __ballot(!have && !out), busy = __ballot(have);
            if (!(idle | busy)) break;
            if (idle && ((uint32_t)__popcll(idle) >= g.refill || !busy)) {
                if (!have && !out) {
                    i = atomicAdd(&sh_next[0], 1u);
                    if (i >= i1) {
                        out = true;
                    } else {
                        const uint32_t bb = ub[i], d0 = ua[i];
                        if ((bb & 0x8000u) && d0) {              // (d0 = 0: failed in pass A0, ub holds the reason)
                            p0 = cpos[i];
                            pos = p0 + d0;
                            b = bb & 0x7FFFu;
                            have = true;
                        }
                    }
                }
            }
            if (!have) continue;
            uint32_t run;
            const uint32_t len = cds0(pos, run);
            bool fail = len == 0u, done = false;
            uint32_t n = 1;
            if (!fail && run) {
                n = spec_run_blocks(c, len - c.id_len - 1u, b);
                fail = !n || n > bend - b;
            }
            if (!fail) {
                pos += len;
                b += n;
                if (b >= bend || (pos < s.limit && s2_marked(marks, pos))) done = true;
                else if (pos >= s.limit) fail = true;
            }
            if (done && pos - p0 > 0xFFFFu) fail = true;
            if (fail || done) {
                // (a walk that fails this close to the end of the window, or is longer than a record holds, has
                // probably run out of the window: kS2Limited)
                ua[i] = fail ? (uint16_t)0 : (uint16_t)(pos - p0);
                ub[i] = fail ? (uint16_t)((near_end(pos) || done) ? kS2Limited : 0u) : (uint16_t)b;
                have = false;
            }
        }
    }
    __syncthreads();
    stamp(7);
    // B. table steps (aec_spec2.h: s2_table_step, tables linked by candidate index) from where pass A left the walk
    {
        const uint32_t bend = c.rsi;
        uint32_t i = 0, idx = 0, b = 0;
        bool have = false, out = false;
        while (true) {
            const uint64_t idle = __ballot(!have && !out), busy = __ballot(have);
            if (!(idle | busy)) break;
            if (idle && ((uint32_t)__popcll(idle) >= g.refill || !busy)) {
                if (!have && !out) {
                    i = atomicAdd(&sh_next[1], 1u);
                    if (i >= i1) {
                        out = true;
                    } else {
                        const uint32_t d = ua[i];
                        b = ub[i];
                        if (d && b < bend) {
                            idx = s2_index(w, (uint32_t)cpos[i] + d);
                            if (idx == kS2NoIndex) { ua[i] = 0; ub[i] = 0; }   // (cannot happen: pass A stopped on a marked boundary)
                            else have = true;
                        }
                    }
                }
            }
            if (!have) continue;
            uint32_t end = 0;
            const uint32_t st = s2_table_step(w, c, idx, b, bend, end);
            if (st != 1u) {
                const uint32_t a = st == 2u ? end - cpos[i] : 0u;
                ua[i] = a > 0xFFFFu ? (uint16_t)0 : (uint16_t)a;
                ub[i] = (st == 0u || (st == 2u && a > 0xFFFFu)) ? (uint16_t)kS2Limited : (uint16_t)0;   // (read only where ua is 0)
                have = false;
            }
        }
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
        // (y = 1: no chain -- a count of 0 -- and the note for k_bridge that the hypothesis ran out of the window)
        const bool limited = !cnt && !ua[i] && ub[i] == kS2Limited;
        grec[at] = make_uint2(ua[i], cnt ? ((cnt << 24) | (pos - cpos[i])) : (limited ? 1u : 0u));
        if (limited && blist) {                               // k_bridge's work list
            // (bit 31: the hypothesis starts with an UNCOMPRESSED coded data set -- what an RSI of incompressible
            // data does, and one in 2^id_len of the wrong phases that run out of their window on a clean stream:
            // k_bridge takes those from the first span on, the others once the walker has met enough of their kind)
            const uint32_t q0 = cpos[i];
            const uint64_t two = ((uint64_t)win[q0 >> 5] << 32) | win[(q0 >> 5) + 1u];
            const uint32_t id = (uint32_t)((two << (q0 & 31u)) >> (64u - c.id_len));
            const uint32_t slot = atomicAdd(blist_cnt, 1u);
            // (bits 26 .. 30: the header itself)
            if (slot < kS2BridgeCap) blist[slot] = (uint32_t)at | (id << 26) | (id == (1u << c.id_len) - 1u ? 0x80000000u : 0u);
        }
        gcpos[at] = (uint16_t)(cpos[i] - c0);
    }
    for (uint32_t i = tid; i < cw; i += nt) {
        gbitmap[gw0 + i] = marks[(c0 >> 5) + i];
        gpre[gw0 + i] = (uint16_t)(mpre[(c0 >> 5) + i] - i0);
    }
    if (tid == 0) gccnt[blockIdx.x] = ncore;
    stamp(6);
}

// ---- the same tables from a table of EVERY position's coded data set (round 5) --------------------------------
// k_spec2 parses a coded data set wherever a walk stands (~100 vector instructions per parse, spec_cds_fast): the sync
// chains alone take 27 000 parses per window, the catch-up of the RSI hypotheses 15 000 more, and the two phases were
// 220 k of the window's 320 k cycles.  Here the length of the coded data set that would begin at a bit is tabulated for
// ALL bits of the window first -- F[q], the nxt[] entry format of aec_spec.h -- which costs ~30 instructions per bit and
// no divergence, because the end of a unary part is ONE lookup once the positions of the window's 1-bits lie in a
// table: the n-th 1-bit behind q1 is ones[rank(q1) + n - 1].  (The table of ones is built tile by tile in the space
// the candidate arrays take later; rank is a prefix count per word + one popcount.)  Every step of a chain or of a
// catch-up walk is then one LDS read.  Everything else -- marks, candidates, hop tables, hypotheses, records -- is
// k_spec2's, in the same order and with the same results: the tables a span gets are bit for bit the ones k_spec2
// writes (tests: AEC_S2_V4=0/1 in the tuning build give identical index results; k_spec_verify on both).
// LDS: win | marks | rank | sel | mpre | [cpos cnxt ua ub] (the ones tile while F is filled) | F[W] (later chop4
// chop16 csucc).  W = 48 kbit fits the 160 KB of a CU.
__device__ __forceinline__ void
spec4_window(const uint32_t wid, const Cfg &c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
             uint64_t tab_lo, uint64_t start_bit, const Spec2Geom &g, uint32_t *__restrict__ gbitmap,
             uint16_t *__restrict__ gpre, uint2 *__restrict__ grec, uint16_t *__restrict__ gcpos,
             uint32_t *__restrict__ gccnt, unsigned long long *__restrict__ prof, const uint64_t *__restrict__ starts,
             uint32_t nstarts, uint32_t *__restrict__ blist, uint32_t *__restrict__ blist_cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t spec_lds[];
    __shared__ uint32_t sh_total, sh_next[2], sh_part[16];
    auto stamp = [&](int k) {
        if (prof && threadIdx.x == 0) prof[(size_t)wid * 16 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const uint32_t W = g.lead + g.core + g.look, nw = W / 32u, cap = g.cap_lds;
    uint32_t *win = spec_lds;
    uint32_t *marks = win + nw + 4;
    uint16_t *rank = reinterpret_cast<uint16_t *>(marks + nw);
    uint16_t *sel = rank + nw + 2;
    uint16_t *mpre = sel + nw + 2;
    uint16_t *cpos = mpre + nw + 2;
    uint16_t *cnxt = cpos + cap;
    uint16_t *ua = cnxt + cap;
    uint16_t *ub = ua + cap;
    uint16_t *F = ub + cap;                           // W entries; dead once the catch-up walks are through
    uint16_t *chop4 = F;
    uint16_t *chop16 = chop4 + cap;
    uint16_t *csucc = chop16 + cap;
    uint16_t *ones = cpos;                            // 4 * cap entries, while F is filled
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint64_t core_abs = tab_lo + (uint64_t)wid * g.core;
    const uint64_t gw0 = ((uint64_t)wid * g.core) >> 5;
    const uint32_t cw = g.core / 32u;
    auto give_up = [&]() {
        for (uint32_t i = tid; i < cw; i += nt) {
            gbitmap[gw0 + i] = 0u;
            gpre[gw0 + i] = 0;
        }
        if (tid == 0) gccnt[wid] = 0u;
    };
    if (core_abs >= end_bit) {
        give_up();
        return;
    }
    const uint64_t wstart = core_abs >= g.lead ? core_abs - g.lead : 0;
    const uint32_t c0 = (uint32_t)(core_abs - wstart), c1 = c0 + g.core;
    const uint64_t w0 = wstart >> 5;
    for (uint32_t i = tid; i < nw + 4; i += nt) {
        const uint64_t idx = w0 + i;
        win[i] = idx < nwords ? bswap32(words[idx]) : 0u;
        if (i < nw) marks[i] = 0u;
    }
    __syncthreads();
    auto prefix16 = [&](const uint32_t *src, uint16_t *dst) {
        const uint32_t nwv = nt >> 6, wv = tid >> 6, ln = tid & 63u;
        const uint32_t per = ((nw + nwv - 1u) / nwv + 63u) & ~63u;
        const uint32_t lo = wv * per, hi = lo + per < nw ? lo + per : nw;
        uint32_t carry = 0;
        for (uint32_t base = lo; base < hi; base += 64) {
            const uint32_t i = base + ln;
            const uint32_t pc = i < hi ? (uint32_t)__popc(src[i]) : 0u;
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < hi) dst[i + 1] = (uint16_t)(carry + incl);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (ln == 0) sh_part[wv] = carry;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t v = 0; v < nwv; v++) {
            const uint32_t t = sh_part[v];
            before += v < wv ? t : 0u;
            all += t;
        }
        for (uint32_t i = lo + ln; i < hi; i += 64) dst[i + 1] = (uint16_t)(dst[i + 1] + before);
        if (tid == 0) {
            dst[0] = 0;
            sh_total = all;
        }
        __syncthreads();
    };
    prefix16(win, rank);
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += nt) {
        const uint32_t lo = rank[i], hi = rank[i + 1], m = (lo + 31u) >> 5;
        if (32u * m + 1u > lo && 32u * m + 1u <= hi) sel[m] = (uint16_t)i;
    }
    __syncthreads();
    const uint64_t left = end_bit - wstart;
    const SpecWin s{win, rank, sel, nw, left < W ? (uint32_t)left : W};
    stamp(1);
    // ---- 0. F[q] for every bit of the window
    {
        const uint32_t il = c.id_len, idmax = (1u << il) - 1u, unc_len = il + c.bs * c.bps;
        // a tile: `T` positions whose entries are computed + `look` bits behind them, in which any coded data set of a
        // reference encoder ends (longer ones -- a foreign encoder's -- take the search over the rank table)
        const uint32_t tcap = 4u * cap;
        uint32_t look = (il + 1u + c.bs * c.bps + 63u) & ~31u;
        if (look > tcap / 2u) look = (tcap / 2u) & ~31u;
        const uint32_t T = (tcap - look) & ~31u;
        for (uint32_t t0 = 0; t0 < s.limit; t0 += T) {
            const uint32_t wlo = t0 >> 5, whi = (t0 + T + look) >> 5 < nw ? (t0 + T + look) >> 5 : nw;
            const uint32_t r0 = rank[wlo], tcnt = (uint32_t)rank[whi] - r0;
            // positions just behind the 1-bits of words [wlo, whi), one lane per byte
            for (uint32_t j = tid; j < (whi - wlo) * 4u; j += nt) {
                const uint32_t wi = wlo + (j >> 2), k = j & 3u, word = win[wi];
                uint32_t at = (uint32_t)rank[wi] - r0 + (k ? (uint32_t)__popc(word >> (32u - 8u * k)) : 0u);
                uint32_t bits = (word << (8u * k)) & 0xFF000000u;
                const uint32_t base = wi * 32u + 8u * k + 1u;
                while (bits) {
                    const uint32_t z = (uint32_t)__builtin_clz(bits);
                    bits &= ~(0x80000000u >> z);
                    ones[at++] = (uint16_t)(base + z);
                }
            }
            __syncthreads();
            const uint32_t t1 = t0 + T < s.limit ? t0 + T : s.limit;
            // (one position per lane and round: consecutive lanes read and write consecutive LDS addresses.  Half a word --
            // 16 positions -- per lane with the rank carried along bit by bit was measured SLOWER, 66 k against 59 k cycles
            // per window: the lanes' table reads and stores then lie 32 bytes apart, eight to a bank)
            for (uint32_t q = t0 + tid; q < t1; q += nt) {
                const uint32_t w = q >> 5, sh = q & 31u;
                const uint32_t a = win[w], b = win[w + 1u];
                const uint32_t h = sh ? __builtin_amdgcn_alignbit(a, b, 32u - sh) : a;
                const uint32_t id = h >> (32u - il);
                const bool unc = id == idmax, low = id == 0u;
                const uint32_t selbit = (h >> (31u - il)) & 1u;
                const uint32_t off1 = il + (low ? 1u : 0u);
                const uint32_t n = low ? (selbit ? c.bs / 2u : 1u) : c.bs;
                const uint32_t add = low ? 0u : n * (id - 1u);
                const uint32_t q1 = q + off1, sh1 = q1 & 31u;
                const uint32_t w1 = (q1 >> 5) == w ? a : b;
                const uint32_t r1 = (uint32_t)rank[q1 >> 5] + (sh1 ? (uint32_t)__popc(w1 >> (32u - sh1)) : 0u);
                const uint32_t k = r1 + n - 1u - r0;                       // index of the 1-bit that ends the unary part
                uint32_t e = k < tcnt ? (uint32_t)ones[k] : 0u;
                const bool in = q + il + 1u <= s.limit;
                if (in && !unc && q1 < s.limit && k >= tcnt) e = spec_select(s, r1 + n);   // (beyond the tile: rare)
                const uint32_t end = unc ? q + unc_len : e + add;
                const bool ok = in && (unc || (q1 < s.limit && e != 0u && e != kSpecInvalid)) && end <= s.limit && end - q < 4096u;
                F[q] = ok ? (uint16_t)((end - q) | ((low && !selbit) ? kNxtZero : kNxtBlock)) : (uint16_t)0;
            }
            __syncthreads();
        }
    }
    stamp(2);
    const bool use_fast = g.fast != 0;
    auto near_end = [&](uint32_t pos) { return pos + 4096u > s.limit; };
    auto cds1 = [&](uint32_t q, uint32_t &run) -> uint32_t {
        if (!use_fast) return spec_cds(s, c, q, 1u, run);
        uint32_t len = spec_cds_fast<1>(win, s.limit, c, q, run);
        if (len == kSpecUnresolved) len = spec_cds(s, c, q, 1u, run);
        return len;
    };
    // (q < s.limit; `run` != 0: a zero-block coded data set, the callers take its code from the length)
    auto cds0 = [&](uint32_t q, uint32_t &run) -> uint32_t {
        const uint32_t e = F[q];
        run = e & kNxtZero;
        return e & 0xFFFu;
    };

    // ---- 1. sync chains: burn in, then mark until a marked boundary is met
    if (tid == 0 && start_bit >= wstart && start_bit - wstart < s.limit) {
        const uint32_t q = (uint32_t)(start_bit - wstart);
        atomicOr(&marks[q >> 5], 1u << (31u - (q & 31u)));
    }
    if (starts && tid == 0) {
        uint32_t lo = 0, hi = nstarts;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (starts[mid] * 8u < wstart) lo = mid + 1u;
            else hi = mid;
        }
        for (; lo < nstarts && starts[lo] * 8u - wstart < s.limit; lo++) {
            const uint32_t q = (uint32_t)(starts[lo] * 8u - wstart);
            atomicOr(&marks[q >> 5], 1u << (31u - (q & 31u)));
        }
    }
    for (uint32_t q0 = tid * g.stride; q0 < s.limit; q0 += nt * g.stride) {
        uint32_t q = q0;
        bool ok = true;
        for (uint32_t k = 0; k < g.burn && ok; k++) {
            const uint32_t len = q < s.limit ? (uint32_t)F[q] & 0xFFFu : 0u;
            ok = len != 0;
            q += len;
        }
        while (ok && q < s.limit) {
            const uint32_t bit = 1u << (31u - (q & 31u));
            if (atomicOr(&marks[q >> 5], bit) & bit) break;
            const uint32_t len = (uint32_t)F[q] & 0xFFFu;
            ok = len != 0;
            q += len;
        }
    }
    // ---- 1b. runs of uncompressed coded data sets (k_spec2, step 1b)
    {
        const uint32_t il = c.id_len, step = il + c.bs * c.bps;
        auto ones_of = [&](uint32_t a, uint32_t b, uint32_t d, uint32_t sh) -> uint32_t {
            const uint64_t hi = ((uint64_t)a << 32) | b;
            const uint64_t y = sh ? (hi << sh) | ((uint64_t)d >> (32u - sh)) : hi;
            uint64_t m = y;
            for (uint32_t k = 1; k < il; k++) m &= y << k;
            return (uint32_t)(m >> 32);
        };
        const uint32_t w1 = step >> 5, s1 = step & 31u, w2 = (2u * step) >> 5, s2 = (2u * step) & 31u,
                       w3 = (3u * step) >> 5, s3 = (3u * step) & 31u;
        for (uint32_t i = tid; i < nw && g.uncrun; i += nt) {
            if (i * 32u + 32u + 7u * step + il > s.limit) break;
            const uint32_t a0 = win[i], b0 = win[i + 1u];
            const uint32_t a1 = win[i + w1], b1 = win[i + w1 + 1u], d1 = win[i + w1 + 2u];
            const uint32_t a2 = win[i + w2], b2 = win[i + w2 + 1u], d2 = win[i + w2 + 2u];
            const uint32_t a3 = win[i + w3], b3 = win[i + w3 + 1u], d3 = win[i + w3 + 2u];
            uint32_t hits = ones_of(a0, b0, 0u, 0u) & ones_of(a1, b1, d1, s1) & ones_of(a2, b2, d2, s2) & ones_of(a3, b3, d3, s3);
            for (uint32_t k = 4; k < 8u && hits; k++) {
                const uint32_t wk = i + ((k * step) >> 5);
                hits &= ones_of(win[wk], win[wk + 1u], win[wk + 2u], (k * step) & 31u);
            }
            while (hits) {
                const uint32_t j = (uint32_t)__builtin_clz(hits);
                hits &= ~(0x80000000u >> j);
                uint32_t q = i * 32u + j;
                bool ok = true;
                while (ok && q < s.limit) {
                    const uint32_t bit = 1u << (31u - (q & 31u));
                    if (atomicOr(&marks[q >> 5], bit) & bit) break;
                    const uint32_t len = (uint32_t)F[q] & 0xFFFu;
                    ok = len != 0;
                    q += len;
                }
            }
        }
    }
    __syncthreads();
    stamp(3);
    prefix16(marks, mpre);
    __syncthreads();
    const uint32_t ncand = sh_total;
    if (ncand > cap) {
        give_up();
        return;
    }
    // ---- 2. the candidates' positions and coded data sets
    for (uint32_t i = tid; i < nw; i += nt) {
        uint32_t m = marks[i], idx = mpre[i];
        while (m) {
            const uint32_t b = (uint32_t)__builtin_clz(m);
            m &= ~(0x80000000u >> b);
            const uint32_t q = i * 32u + b;
            cpos[idx] = (uint16_t)q;
            cnxt[idx] = q < s.limit ? F[q] : (uint16_t)0;
            idx++;
        }
    }
    __syncthreads();
    stamp(8);
    // ---- 3. the RSI hypothesis at every candidate of the core
    const uint32_t i0 = mpre[c0 >> 5], i1 = mpre[c1 >> 5 < nw ? c1 >> 5 : nw];
    const uint32_t ncore = i1 - i0;
    if (tid == 0) sh_next[0] = sh_next[1] = i0;
    // A0. the first coded data set of every hypothesis (with the reference sample)
    {
        const uint32_t ref_first = (c.flags & F_PREPROCESS) ? 1u : 0u, bend = c.rsi;
        for (uint32_t i = i0 + tid; i < i1; i += nt) {
            const uint32_t p0 = cpos[i];
            uint32_t delta = 0, bb = near_end(p0) ? kS2Limited : 0u;
            if (p0 < s.limit) {
                uint32_t run;
                const uint32_t len = ref_first ? cds1(p0, run) : cds0(p0, run);
                bool fail = len == 0u;
                uint32_t n = 1;
                if (!fail && run) {
                    n = spec_run_blocks(c, len - c.id_len - 1u - (ref_first ? c.bps : 0u), 0u);
                    fail = !n || n > bend;
                    if (fail) bb = 0;
                }
                if (!fail) {
                    const uint32_t pos = p0 + len;
                    bb = pos >= s.limit ? kS2Limited : 0u;
                    if (n >= bend || (pos < s.limit && s2_marked(marks, pos))) {
                        delta = len;
                        bb = n;
                    } else if (pos < s.limit) {
                        delta = len;
                        bb = n | 0x8000u;
                    }
                }
            }
            ua[i] = (uint16_t)delta;
            ub[i] = (uint16_t)bb;
        }
    }
    __syncthreads();
    stamp(9);
    // A1. catch-up: table steps (no reference sample) until the walk stands on a marked boundary
    {
        const uint32_t bend = c.rsi;
        uint32_t i = 0, p0 = 0, pos = 0, b = 0;
        bool have = false, out = false;
        while (true) {
            const uint64_t idle = __ballot(!have && !out), busy = __ballot(have);
            if (!(idle | busy)) break;
            if (idle && ((uint32_t)__popcll(idle) >= g.refill || !busy)) {
                // (ONE atomic per wavefront: 1400 of them on one LDS word, lane by lane, were a third of this pass)
                const uint32_t nreq = (uint32_t)__popcll(idle);
                uint32_t base_i = 0;
                if ((tid & 63u) == (uint32_t)__builtin_ctzll(idle)) base_i = atomicAdd(&sh_next[0], nreq);
                base_i = (uint32_t)__builtin_amdgcn_readlane((int)base_i, (int)__builtin_ctzll(idle));
                if (!have && !out) {
                    i = base_i + (uint32_t)__popcll(idle & ((1ull << (tid & 63u)) - 1ull));
                    if (i >= i1) {
                        out = true;
                    } else {
                        const uint32_t bb = ub[i], d0 = ua[i];
                        if ((bb & 0x8000u) && d0) {
                            p0 = cpos[i];
                            pos = p0 + d0;
                            b = bb & 0x7FFFu;
                            have = true;
                        }
                    }
                }
            }
            // A step is one read of the table; the marks are looked at once per group of four plain steps (coded data sets
            // of one block): a walk that has reached the marked chain stays on it, so it may hand over a few boundaries
            // further on -- the table steps behind get to the same end.  Zero runs and failures one at a time.
#pragma unroll 1
            for (uint32_t t = 0; t < 4u; t++) {
                if (have) {
                    uint32_t e = F[pos];
#pragma unroll 1
                    for (uint32_t u = 0; u < 4u && (e >> 12) == (kNxtBlock >> 12); u++) {
                        pos += e & 0xFFFu;
                        b++;
                        if (b >= bend || pos >= s.limit) break;
                        if (u < 3u) e = F[pos];
                    }
                    bool fail = false, done = false;
                    if (b >= bend) {
                        done = true;
                    } else if (pos >= s.limit) {
                        fail = true;
                    } else if (s2_marked(marks, pos)) {
                        done = true;
                    } else {
                        e = F[pos];
                        if ((e >> 12) != (kNxtBlock >> 12)) {                 // a zero run, or no coded data set from here
                            const uint32_t len = e & 0xFFFu;
                            uint32_t n = 0;
                            if (len) n = spec_run_blocks(c, len - c.id_len - 1u, b);
                            if (!len || !n || n > bend - b) {
                                fail = true;
                            } else {
                                pos += len;
                                b += n;
                                if (b >= bend || (pos < s.limit && s2_marked(marks, pos))) done = true;
                                else if (pos >= s.limit) fail = true;
                            }
                        }
                    }
                    if (done && pos - p0 > 0xFFFFu) fail = true;
                    if (fail || done) {
                        ua[i] = fail ? (uint16_t)0 : (uint16_t)(pos - p0);
                        ub[i] = fail ? (uint16_t)((near_end(pos) || done) ? kS2Limited : 0u) : (uint16_t)b;
                        have = false;
                    }
                }
                if (!__any(have)) break;
            }
        }
    }
    __syncthreads();
    stamp(4);
    // ---- (F is dead: the hop tables take its place)
    S2Win w{s, marks, mpre, cnxt, csucc, chop4, chop16, cpos, ncand};
    for (uint32_t i = tid; i < ncand; i += nt) csucc[i] = s2_succ(w, i);
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) chop4[i] = s2_hop4(w, c, i);
    __syncthreads();
    for (uint32_t i = tid; i < ncand; i += nt) chop16[i] = s2_hop16(w, i);
    __syncthreads();
    stamp(5);
    // B. table steps from where pass A left the walk
    {
        const uint32_t bend = c.rsi;
        uint32_t i = 0, idx = 0, b = 0;
        bool have = false, out = false;
        while (true) {
            const uint64_t idle = __ballot(!have && !out), busy = __ballot(have);
            if (!(idle | busy)) break;
            if (idle && ((uint32_t)__popcll(idle) >= g.refill || !busy)) {
                const uint32_t nreq = (uint32_t)__popcll(idle);
                uint32_t base_i = 0;
                if ((tid & 63u) == (uint32_t)__builtin_ctzll(idle)) base_i = atomicAdd(&sh_next[1], nreq);
                base_i = (uint32_t)__builtin_amdgcn_readlane((int)base_i, (int)__builtin_ctzll(idle));
                if (!have && !out) {
                    i = base_i + (uint32_t)__popcll(idle & ((1ull << (tid & 63u)) - 1ull));
                    if (i >= i1) {
                        out = true;
                    } else {
                        const uint32_t d = ua[i];
                        b = ub[i];
                        if (d && b < bend) {
                            idx = s2_index(w, (uint32_t)cpos[i] + d);
                            if (idx == kS2NoIndex) { ua[i] = 0; ub[i] = 0; }
                            else have = true;
                        }
                    }
                }
            }
#pragma unroll 1
            for (uint32_t t = 0; t < 4u; t++) {
                if (have) {
                    uint32_t end = 0;
                    const uint32_t st = s2_table_step(w, c, idx, b, bend, end);
                    if (st != 1u) {
                        const uint32_t a = st == 2u ? end - cpos[i] : 0u;
                        ua[i] = a > 0xFFFFu ? (uint16_t)0 : (uint16_t)a;
                        ub[i] = (st == 0u || (st == 2u && a > 0xFFFFu)) ? (uint16_t)kS2Limited : (uint16_t)0;
                        have = false;
                    }
                }
                if (!__any(have)) break;
            }
        }
    }
    __syncthreads();
    stamp(6);
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
        const uint64_t at = (uint64_t)wid * g.cap_core + (i - i0);
        const bool limited = !cnt && !ua[i] && ub[i] == kS2Limited;
        grec[at] = make_uint2(ua[i], cnt ? ((cnt << 24) | (pos - cpos[i])) : (limited ? 1u : 0u));
        if (limited && blist) {
            const uint32_t q0 = cpos[i];
            const uint64_t two = ((uint64_t)win[q0 >> 5] << 32) | win[(q0 >> 5) + 1u];
            const uint32_t id = (uint32_t)((two << (q0 & 31u)) >> (64u - c.id_len));
            const uint32_t slot = atomicAdd(blist_cnt, 1u);
            if (slot < kS2BridgeCap) blist[slot] = (uint32_t)at | (id << 26) | (id == (1u << c.id_len) - 1u ? 0x80000000u : 0u);
        }
        gcpos[at] = (uint16_t)(cpos[i] - c0);
    }
    for (uint32_t i = tid; i < cw; i += nt) {
        gbitmap[gw0 + i] = marks[(c0 >> 5) + i];
        gpre[gw0 + i] = (uint16_t)(mpre[(c0 >> 5) + i] - i0);
    }
    if (tid == 0) gccnt[wid] = ncore;
    stamp(7);
}

__global__ void __launch_bounds__(1024)
k_spec4(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, uint64_t tab_lo,
        uint64_t start_bit, const Spec2Geom g, uint32_t *__restrict__ gbitmap, uint16_t *__restrict__ gpre,
        uint2 *__restrict__ grec, uint16_t *__restrict__ gcpos, uint32_t *__restrict__ gccnt,
        unsigned long long *__restrict__ prof, const uint64_t *__restrict__ starts = nullptr, uint32_t nstarts = 0,
        uint32_t *__restrict__ blist = nullptr, uint32_t *__restrict__ blist_cnt = nullptr,
        const uint32_t *__restrict__ wlist = nullptr, const uint32_t *__restrict__ wcount = nullptr,
        const uint32_t *__restrict__ only_where_zero = nullptr, uint32_t other_core = 0, uint32_t other_nwin = 0)
{
    if (g.skip_if && *g.skip_if) return;
    // dense mode of a few windows (small inputs): every window looks itself whether an ordinary window over its core
    // gave up -- one launch instead of the list's two
    if (only_where_zero) {
        const uint64_t a = (uint64_t)blockIdx.x * g.core, b = a + g.core - 1u;
        const uint32_t j0 = (uint32_t)(a / other_core), j1 = (uint32_t)(b / other_core);
        bool any = false;
        for (uint32_t j = j0; j <= j1 && j < other_nwin; j++) any = any || only_where_zero[j] == 0u;
        if (!any) {
            if (threadIdx.x == 0) gccnt[blockIdx.x] = 0u;
            return;
        }
    }
    if (!wlist) {
        spec4_window(blockIdx.x, c, words, nwords, end_bit, tab_lo, start_bit, g, gbitmap, gpre, grec, gcpos, gccnt, prof, starts,
                     nstarts, blist, blist_cnt);
        return;
    }
    // dense mode: the windows on the list k_dense_pick left (those over an ordinary window that gave up), a few per
    // workgroup -- on nearly every stream the list is empty and the launch is a few hundred workgroups that return
    const uint32_t n = *wcount;
    for (uint32_t li = blockIdx.x; li < n; li += gridDim.x) {
        spec4_window(wlist[li], c, words, nwords, end_bit, tab_lo, start_bit, g, gbitmap, gpre, grec, gcpos, gccnt, prof, starts,
                     nstarts, blist, blist_cnt);
        __syncthreads();
    }
}

// which windows of the dense tables are to be built: those whose core lies over an ordinary window with a count of 0
__global__ void __launch_bounds__(256)
k_dense_pick(const uint32_t *__restrict__ sparse_ccnt, uint32_t sparse_core, uint32_t sparse_nwin, uint32_t dcore, uint32_t dnwin,
             uint32_t *__restrict__ dccnt, uint32_t *__restrict__ wlist, uint32_t *__restrict__ wcount,
             const uint32_t *__restrict__ skip_if = nullptr)
{
    if (skip_if && *skip_if) return;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= dnwin) return;
    const uint64_t a = (uint64_t)w * dcore, b = a + dcore - 1u;
    const uint32_t j0 = (uint32_t)(a / sparse_core), j1 = (uint32_t)(b / sparse_core);
    bool any = false;
    for (uint32_t j = j0; j <= j1 && j < sparse_nwin; j++) any = any || sparse_ccnt[j] == 0u;
    if (any) wlist[atomicAdd(wcount, 1u)] = w;
    else dccnt[w] = 0u;
}

#ifdef AEC_TUNING
// (diagnostics) every record of the window tables against the plain parse of its RSI from device memory
__global__ void k_spec_verify(const Cfg c, const TrStream s, const SparseTables t, uint32_t nwin, uint32_t *bad)
{
    const uint32_t w = blockIdx.x;
    if (w >= nwin) return;
    for (uint32_t i = threadIdx.x; i < t.ccnt[w]; i += blockDim.x) {
        const uint2 rec = t.rec[(uint64_t)w * t.cap + i];
        const uint64_t p = t.lo + (uint64_t)w * t.core + t.cpos[(uint64_t)w * t.cap + i];
        // the position must be marked and have this index
        uint2 r2;
        uint32_t wv, ix;
        if (!sparse_lookup(t, p, r2, wv, ix) || wv != w || ix != i) {
            if (atomicAdd(&bad[0], 1u) < 4u) printf("window %u cand %u pos %llu: lookup says window %u index %u\n", w, i, (unsigned long long)p, wv, ix);
            continue;
        }
        if (!rec.x) continue;
        uint64_t q = p;
        uint32_t b = 0;
        bool ok = true;
        while (b < c.rsi && ok) {
            uint32_t nz;
            const uint32_t len = tr_cds(s, c, q, (b == 0 && (c.flags & F_PREPROCESS)) ? 1u : 0u, nz);
            const uint32_t nb = len ? tr_blocks(c, nz, b) : 0u;
            if (!len || !nb || nb > c.rsi - b) ok = false;
            q += len;
            b += nb;
        }
        if (!ok || q - p != rec.x) {
            if (atomicAdd(&bad[1], 1u) < 8u)
                printf("window %u cand %u pos %llu: record %u, parse %llu (%s)\n", w, i, (unsigned long long)p, rec.x,
                       (unsigned long long)(q - p), ok ? "ok" : "fails");
        }
    }
}
#endif

// RSIs the window tables could not resolve because the walk ran out of its window -- longer than the look-ahead:
// incompressible stretches in an otherwise low-entropy stream -- parsed coded data set by coded data set from the
// stream where it lies, one lane per hypothesis on the list k_spec2 leaves (all of them around such a stretch, the
// true RSI start among them; empty on well-behaved streams).  Without it every such RSI costs the walker a serial
// walk AND the chunk-level lookup of its 256 windows: 1 in 1000 RSIs of this kind doubled the index time.
// (16 bytes per memory round trip instead of 4: the parse below waits for nothing else)
struct QuadFetch {
    const uint32_t *words;
    uint64_t nwords;
    mutable uint64_t have = ~0ull;
    mutable uint32_t q[4] = {0, 0, 0, 0};
    __device__ uint32_t operator()(uint64_t idx) const
    {
        if ((idx >> 2) != have) {
            have = idx >> 2;
            const uint64_t b = have << 2;
            if (b + 4 <= nwords) {
                struct __attribute__((packed, aligned(4))) Q { uint32_t a, b, c, d; };
                const Q v = *reinterpret_cast<const Q *>(words + b);
                q[0] = v.a; q[1] = v.b; q[2] = v.c; q[3] = v.d;
            } else {
                for (uint32_t k = 0; k < 4; k++) q[k] = b + k < nwords ? words[b + k] : 0u;
            }
        }
        return bswap32(q[idx & 3u]);
    }
};

// unc_only: take the hypothesis only if its RSI looks like incompressible data -- it begins with four uncompressed
// coded data sets and holds at most eight of another kind (random 16-bit samples: one block in 300 comes out a
// little shorter as a split).  The headers of a run of uncompressed coded data sets lie a fixed distance apart, so
// the run is a series of independent reads, eight at a time, and a wrong phase on a clean stream is out after the first.
__device__ void bridge_one(const Cfg &c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
                           const SparseTables &t, uint2 *__restrict__ rec_out, uint32_t at, bool unc_only)
{
    const uint32_t w = at / t.cap;
    if (t.rec[at].x || t.rec[at].y != 1u) return;
    const uint64_t pos = t.lo + (uint64_t)w * t.core + t.cpos[at];
    if (pos >= end_bit) return;
    const uint32_t step = c.id_len + c.bs * c.bps, idmax = (1u << c.id_len) - 1u;
    auto id_at = [&](uint64_t q) -> uint32_t {
        const uint64_t wi = q >> 5;
        const uint64_t two = ((uint64_t)bswap32(words[wi < nwords ? wi : nwords - 1u]) << 32) |
                             bswap32(words[wi + 1u < nwords ? wi + 1u : nwords - 1u]);
        return (uint32_t)((two << (q & 31u)) >> (64u - c.id_len));
    };
    BitReaderT<QuadFetch> br;
    br.init(QuadFetch{words, nwords}, end_bit, pos);
    const bool pp = c.flags & F_PREPROCESS;
    // (an RSI of the reference encoder is at most all blocks uncompressed; whatever is longer is left to the walker)
    const uint64_t longest = (uint64_t)c.rsi * step + c.bps + 64u;
    uint32_t b = 0, slow = 0;
    while (b < c.rsi) {
        if (br.cnt < c.id_len) br.refill();
        const uint32_t id = (uint32_t)(br.win >> (64u - c.id_len));
        if (id == idmax) {
            const uint64_t p = br.pos;
            const uint32_t left = c.rsi - b;
            uint32_t ids[7];
#pragma unroll
            for (uint32_t k = 0; k < 7u; k++) ids[k] = k + 1u < left ? id_at(p + (uint64_t)(k + 1u) * step) : 0u;
            uint32_t lead = 1;
            bool all = true;
#pragma unroll
            for (uint32_t k = 0; k < 7u; k++) {
                all = all && ids[k] == idmax;
                lead += all ? 1u : 0u;
            }
            if (unc_only && b == 0 && lead < (c.rsi < 4u ? c.rsi : 4u)) return;
            if (p + (uint64_t)lead * step > end_bit) return;
            br.skip((uint64_t)lead * step);
            b += lead;
        } else {
            if (unc_only && (b == 0 || ++slow > 8u)) return;
            uint32_t nblk = 1;
            if (skip_cds(br, c, (pp && b == 0) ? 1u : 0u, b, nblk) != DEC_OK) return;
            b += nblk;
        }
        if (br.pos - pos > longest) return;
    }
    if (b != c.rsi || br.pos > end_bit || br.pos - pos > 0xFFFFFFFFull) return;
    const uint64_t e1 = br.pos;
    uint32_t y = 1u;
    // Is the RSI start behind it a candidate of ITS window?  A window that begins inside a long RSI may have no chain
    // on the boundaries yet where the RSI ends (burn-in), and the walker would have to take the next RSI coded data
    // set by coded data set: that RSI goes into the record as well then -- a chain of two, the walker lands on the
    // third, a whole ordinary RSI further on.
    if (!(c.flags & F_PAD_RSI) && e1 < t.hi && e1 < end_bit) {
        uint2 r2;
        uint32_t wv, ix;
        if (!sparse_lookup(t, e1, r2, wv, ix)) {
            b = 0;
            bool ok = true;
            while (ok && b < c.rsi) {
                uint32_t nblk = 1;
                ok = skip_cds(br, c, (pp && b == 0) ? 1u : 0u, b, nblk) == DEC_OK && br.pos - e1 <= longest;
                b += nblk;
            }
            if (ok && b == c.rsi && br.pos <= end_bit && br.pos - pos < (1ull << 24)) y = (2u << 24) | (uint32_t)(br.pos - pos);
        }
    }
    rec_out[at] = make_uint2((uint32_t)(e1 - pos), y);
}

__global__ void __launch_bounds__(64)
k_bridge(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit, const SparseTables t,
         uint2 *__restrict__ rec_out, const uint32_t *__restrict__ blist, const uint32_t *__restrict__ blist_cnt,
         const IdxCarry *__restrict__ carry, uint32_t first_span, uint32_t min_serial)
{
    if (t.skip_if && *t.skip_if) return;
    // Only for streams that need it: well-behaved streams have a dozen such hypotheses per window too (wrong phases
    // whose zero runs come out long), each a parse of several average RSIs for nothing.  The walker counts the
    // RSIs it had to walk itself; from kS2BridgeAfter of them on the rest of the stream is bridged.
    // Hypotheses that start with an uncompressed coded data set (bit 31 of their entry) are bridged always: cheap on a
    // clean stream (few), and what the true start of an incompressible RSI looks like.
    const bool all = !first_span && carry->n_serial >= min_serial;
    const uint32_t n = *blist_cnt < kS2BridgeCap ? *blist_cnt : kS2BridgeCap;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t e = blist[j];
        if (all || (e >> 31)) bridge_one(c, words, nwords, end_bit, t, rec_out, e & 0x03FFFFFFu, !all);
    }
}

// ---- wide walker: every candidate of a chunk's first window chases the window chain through the chunk
__global__ void __launch_bounds__(256)
k_wide(const SparseTables t, uint32_t nwin, uint64_t end_bit, uint4 *__restrict__ wide)
{
    if (t.skip_if && *t.skip_if) return;
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
        if (!sparse_lookup(t, pos, rec, wv, ix)) {
            ok = 0;
            break;
        }
        if (rec.y >> 24) {
            pos += rec.y & 0xFFFFFFu;
            cnt += rec.y >> 24;
        } else if (rec.x && pos + rec.x <= end_bit) {       // no chain from here (k_bridge's records): this RSI alone
            pos += rec.x;
            cnt++;
        } else {
            ok = 0;
            break;
        }
    }
    // (a chain that ends at the end of the input inside the last chunk is resolved as far as it goes:
    // the walker takes over from its exit)
    wide[(uint64_t)chunk * t.cap + i] = make_uint4((uint32_t)pos, (uint32_t)(pos >> 32), cnt, ok && cnt ? 1u : 0u);
}

// the true chain through the chunks the walker skipped: one lane per chunk records the window hops
__global__ void __launch_bounds__(64)
k_rewalk(const SparseTables t, uint32_t nwin, uint32_t nchunks, uint64_t end_bit,
         const ChunkEntry *__restrict__ entry, IdxHop *__restrict__ hops, uint32_t *__restrict__ nhops,
         uint64_t *__restrict__ rsi_off)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= nchunks) return;
    uint32_t n = 0;
    if (entry[chunk].valid) {
        const uint32_t first = chunk * t.wpc;
        const uint32_t last = first + t.wpc < nwin ? first + t.wpc : nwin;
        const uint64_t stop = t.lo + (uint64_t)last * t.core;
        uint64_t pos = entry[chunk].pos, r = entry[chunk].r;
        IdxHop *out = hops + (uint64_t)chunk * t.wpc * 2u;
        while (pos < stop && pos < end_bit) {
            uint2 rec;
            uint32_t wv, ix;
            if (!sparse_lookup(t, pos, rec, wv, ix)) break;                      // (cannot happen: k_wide went through)
            if ((rec.y >> 24) && n < t.wpc * 2u) {
                out[n++] = IdxHop{pos, r, rec.y >> 24, 0u};
                pos += rec.y & 0xFFFFFFu;
                r += rec.y >> 24;
            } else if (rec.x) {                  // a single RSI (no chain, or no room for another hop): written here
                rsi_off[r] = pos;
                pos += rec.x;
                r++;
            } else {
                break;
            }
        }
    }
    nhops[chunk] = n;
}

// RSI starts inside hops, from the sparse records.  lists == 0: the walker's own hop list (count in
// carry->n_hops); lists > 0: the per-chunk lists of k_rewalk (`stride` entries apart, counts in nhops).
__global__ void k_expand2(const SparseTables t, const IdxCarry *__restrict__ carry, const IdxHop *__restrict__ hops,
                          const uint32_t *__restrict__ nhops, uint32_t lists, uint32_t stride,
                          uint64_t *__restrict__ rsi_off, uint64_t rsi_stride = 0)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    IdxHop h;
    if (lists == 0) {
        if (i >= carry->n_hops) return;
        h = hops[i];
    } else {
        const uint32_t list = i / stride, j = i % stride;
        if (list >= lists || j >= nhops[list]) return;
        h = hops[(uint64_t)list * stride + j];
        rsi_off += (uint64_t)list * rsi_stride;            // (batch of streams: every list numbers its RSIs from 0)
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

// ---- serial RSI index -----------------------------------------------------------------------------
// One wavefront per stream.  The walk itself is serial (every lane executes it redundantly on
// wave-uniform values), but the stream is served from a 16 KiB LDS window that all 64 lanes refill
// with coalesced 16-byte loads, so the parser never waits for HBM: a dependent global load per
// refill of the bit window held the first version at ~5 MB/s of compressed input.
#ifdef AEC_TUNING
__device__ int g_dbg_serial = 0;      // (diagnostics, AEC_IDX_STATS=2: k_index names the RSIs it walks itself)
__device__ int g_idx_no_stretch = 0;  // (A/B: AEC_IDX_STRETCH=0 gives the serial walk of round 4)
#endif
constexpr uint32_t kIdxWindowWords = 4096;
constexpr uint32_t kIdxPiece = 4096;                                   // bits of a piece
constexpr uint32_t kIdxPieceWords = kIdxPiece / 32 + 68;               // + the longest coded data set of an encoder (2118 bits)

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
        const uint64_t *__restrict__ chunk_off, IdxHop *__restrict__ hops, uint32_t hop_cap, IdxCarry *carry,
        uint32_t first, uint32_t last, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot,
        const TwTables sp, ChunkEntry *__restrict__ centry, const SparseTables s2,
        uint32_t *__restrict__ batch_nhops = nullptr, uint64_t stop_near = 0,
        const uint32_t *__restrict__ skip_if = nullptr, const SparseTables s2d = SparseTables{},
        uint32_t serial_cap = 0, uint32_t *__restrict__ delivered = nullptr)
{
    if (skip_if && *skip_if) return;                 // (the phase-locked chains have delivered everything: launch_index_locked)
    // serial_cap / delivered (launch_index_sparse, a small stream in one span): the tables resolve most RSIs of the streams
    // they are made for and next to none of others -- very compressible data, runs of zero blocks -- and this walk is
    // then 50 ns per block on ONE wavefront (16 MiB of 12-bit data in blocks of 8: 54 ms).  After serial_cap blocks
    // walked serially the walk gives up: *delivered stays 0 and the every-bit scheme enqueued behind takes the stream
    // (0.9 ms for that one); else *delivered = 1 and that scheme's kernels return at once.
    // stop_near (bits; callers that index a stream piece by piece and have more of it than they hand in): an RSI the
    // tables do not resolve within this many bits of the end of the input is not walked serially -- the tables end
    // there for lack of look-ahead, the caller's next piece resolves it -- the pass ends in front of it
    __shared__ __attribute__((aligned(16))) uint32_t win[kIdxWindowWords];
    // (round 5) rank and positions of the 1-bits of a PIECE of the window (kIdxPiece bits + the look-ahead of one coded
    // data set): with them "the coded data set that would begin at bit q" is ~50 vector instructions without a
    // dependence between lanes, so the 64 lanes parse the coded data sets that WOULD begin at the next 64 bits at once
    // and the walk through those bits is a lane read per coded data set (see the serial walk below)
    __shared__ uint16_t prank[kIdxPieceWords + 2];
    __shared__ uint16_t ones[kIdxPieceWords * 32];
    uint64_t r = 0;
    if (chunk_off) {
        start_bit = chunk_off[blockIdx.x] * 8u;
        end_bit = chunk_off[blockIdx.x + 1] * 8u;
        rsi_off += (uint64_t)blockIdx.x * max_rsi;
        res += blockIdx.x;
        if (hops) hops += (uint64_t)blockIdx.x * hop_cap;        // (batch over the window tables: a hop list per stream)
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
    const uint32_t lane = threadIdx.x;
    if (carry && !chunk_off && lane == 0) carry->r_prev = r;
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

    // ---- the piece tables and the parse of a position from them
    const uint32_t look_words = (maxbits + 31u) / 32u + 2u;
    const uint32_t pwords = kIdxPiece / 32u + (look_words < 68u ? look_words : 68u);
#ifdef AEC_TUNING
    const bool stretch = g_idx_no_stretch == 0 && maxbits <= 2118u;      // (AEC_IDX_STRETCH=0: the walk as it was)
#else
    const bool stretch = maxbits <= 2118u;
#endif
    uint64_t piece_base = ~0ull;       // LDS window the piece was made from
    uint32_t pc0 = 0, tcnt = 0, plim = 0;
    auto build_piece = [&](uint32_t from_bit) {
        pc0 = from_bit & ~31u;
        piece_base = base;
        const uint32_t w0 = pc0 >> 5;
        const uint64_t wbits = (uint64_t)kIdxWindowWords * 32u;
        const uint64_t sbits = end_bit > base * 32u ? end_bit - base * 32u : 0u;
        plim = (uint32_t)(sbits < wbits ? sbits : wbits);                 // bits of the window that are stream
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t i0 = 0; i0 < pwords; i0 += 64u) {
            const uint32_t i = i0 + lane, wi = w0 + i;
            const uint32_t word = (i < pwords && wi < kIdxWindowWords) ? win[wi] : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(word);
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < pwords) prank[i + 1u] = (uint16_t)(carry + incl);
            uint32_t at = carry + incl - pc, bits = word;
            // (positions relative to the piece: the window has 131 072 bits, the table 16 bits per entry -- window
            // positions wrapped in the upper half of every window and no entry there resolved: ADVICE round 5)
            const uint32_t bbase = i * 32u + 1u;
            while (bits) {
                const uint32_t z = (uint32_t)__builtin_clz(bits);
                bits &= ~(0x80000000u >> z);
                ones[at++] = (uint16_t)(bbase + z);
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) prank[0] = 0;
        tcnt = carry;
        __syncthreads();
    };
    // entry (aec_spec.h nxt[] format: length | kind, 0 = not resolved here) of the coded data set that would begin at
    // window bit q, pc0 <= q < pc0 + kIdxPiece; ref: with a reference sample behind its header
    auto fentry = [&](uint32_t q, uint32_t ref) -> uint32_t {
        const uint32_t il = c.id_len, w = q >> 5, sh = q & 31u;
        const uint32_t wa = w < kIdxWindowWords - 1u ? w : kIdxWindowWords - 2u;
        const uint32_t a = win[wa], bw = win[wa + 1u];
        const uint32_t h = (uint32_t)(((((uint64_t)a) << 32) | bw) << sh >> 32);
        const uint32_t id = h >> (32u - il);
        const bool unc = id == idmax, low = id == 0u;
        const uint32_t selb = (h >> (31u - il)) & 1u;
        const uint32_t q1 = q + il + (low ? 1u : 0u) + ((ref && !unc) ? c.bps : 0u);
        const uint32_t n = low ? (selb ? c.bs / 2u : 1u) : c.bs - ref;
        const uint32_t w1 = q1 >> 5, sh1 = q1 & 31u;
        const uint32_t w1c = w1 < kIdxWindowWords ? w1 : kIdxWindowWords - 1u;
        const uint32_t pi = w1 - (pc0 >> 5);
        const uint32_t r1 = (uint32_t)prank[pi < kIdxPieceWords ? pi : kIdxPieceWords] + (sh1 ? (uint32_t)__builtin_popcount(win[w1c] >> (32u - sh1)) : 0u);
        const uint32_t k = r1 + n - 1u;
        const uint32_t e = (k < tcnt && n != 0u) ? pc0 + (uint32_t)ones[k] : 0u;
        const uint32_t add = low ? 0u : n * (id - 1u);
        const uint32_t end = unc ? q + il + c.bs * c.bps : e + add;
        const bool ok = (unc || e != 0u) && q1 < plim && end <= plim && end - q < 4096u;
        return ok ? ((end - q) | ((low && !selb) ? kNxtZero : kNxtBlock)) : 0u;
    };

    BitReaderT<LdsWindowFetch> br;
    br.init(LdsWindowFetch{win, base}, end_bit, start_bit);
    uint64_t good = start_bit;
    bool hopped_far = false;           // `good` moved without the sequential reader: it starts from there again
    uint32_t b = 0, status = DEC_OK, nh = 0;
    uint32_t n_serial = (carry && !first) ? carry->n_serial : 0u, n_lookups = (carry && !first) ? carry->n_lookups : 0u;
    // Resumed walk (streaming callers): start_bit is a CDS boundary inside an RSI that began at
    // rsi_start and of which start_block blocks lie before start_bit.
    uint64_t cur_start = start_bit;          // start of the RSI being walked
    if (first && !chunk_off && start_block) {
        b = start_block;
        cur_start = rsi_start;
        if (lane == 0 && max_rsi) rsi_off[0] = rsi_start;
    }
    if (coop) load_regs(good >> 5);
    for (;;) {
        bool hopped = false;
        if (b == 0) {
            if (r >= max_rsi) {
                // (the caller's bound: the walk ends on the START of RSI max_rsi -- with AEC_PAD_RSI on the byte it begins
                // on, as the other schemes report it: the next batch's walk starts there, and the every-bit scheme parses
                // from exactly where it is told to)
                if ((c.flags & F_PAD_RSI) && (good & 7u) && good > start_bit) good = (good + 7u) & ~7ull;
                break;
            }
            // Fast hops over the speculative tables (k_spec): at an RSI start one lookup gives the
            // end of a chain of whole RSIs leaving the window (Xb/Xc; the RSI starts inside the hop
            // are filled in by k_expand), or of this RSI alone (T).  An entry of 0 = not resolved
            // by the tables: that RSI is walked CDS by CDS below.
            // Sparse tables (k_spec2): at a chunk's first window ONE lookup in the wide walker's table
            // takes the walk across the whole chunk (k_rewalk / k_expand2 fill in what lies inside);
            // else the window's chained hop, else this RSI alone.
            while (s2.bitmap && r < max_rsi && good >= s2.lo && good < s2.hi && good < end_bit) {
                uint2 rec;
                uint32_t wv, ix;
                // (the dense tables, built where a window of the ordinary ones gave up: this RSI alone)
                auto dense_hop = [&]() -> bool {
                    uint2 rd;
                    if (!s2d.bitmap || good < s2d.lo || good >= s2d.hi || !dense_lookup(s2d, good, rd)) return false;
                    if (!rd.x || good + rd.x > end_bit) return false;
                    if (lane == 0) rsi_off[r] = good;
                    good += rd.x;
                    r++;
                    n_lookups++;
                    return true;
                };
                if (!sparse_lookup(s2, good, rec, wv, ix)) {
                    if (dense_hop()) {
                        hopped = true;
                        continue;
                    }
                    break;
                }
                if (s2.wide && (wv % s2.wpc) == 0u) {
                    const uint4 wd = s2.wide[(uint64_t)(wv / s2.wpc) * s2.cap + ix];
                    if (wd.w && r + wd.z <= max_rsi) {
                        if (lane == 0) centry[wv / s2.wpc] = ChunkEntry{good, r, 1u, 0u};
                        good = (uint64_t)wd.x | ((uint64_t)wd.y << 32);
                        r += wd.z;
                        hopped = true;
                        continue;
                    }
                }
                const uint32_t xc = rec.y >> 24, xb = rec.y & 0xFFFFFFu, t = rec.x;
                if (xc && r + xc <= max_rsi && nh < hop_cap && good + xb <= end_bit) {
                    if (lane == 0) hops[nh] = IdxHop{good, r, xc, 0u};
                    nh++;
                    good += xb;
                    r += xc;
                } else if (t && good + t <= end_bit) {
                    if (lane == 0) rsi_off[r] = good;
                    good += t;
                    r++;
                } else if (!dense_hop()) {
                    break;
                }
                hopped = true;
            }
            // Hops over the trunk tables (aec_trunk.h).  A record is keyed by the node where its RSI starts --
            // with AEC_PAD_RSI where the RSI in front of it ENDS, so the lookups come before the alignment.
            // At a chunk's first window ONE lookup in the wide walker's table takes the walk across the whole
            // chunk (k_trewalk fills in what lies inside); else record by record.  No record = not resolved by
            // the tables: that RSI is walked CDS by CDS below.
            // (a record of one RSI names the node it ends on: the next record is then ONE read, not a search for the node
            // at a position -- with RSIs of a megabit the walk is a chain of such reads, 2.4 us each before, 1 us now)
            // (Still one dependent read per RSI, 3.4 us on a table of hundreds of megabytes: the wide table above is
            // what keeps their number down.  Fetching the records around where the next ones should lie ahead of time
            // -- the stride between them is an RSI's blocks give or take a few hundred nodes -- was built and changed
            // nothing.)
            uint32_t next_at = 0;
            while (sp.bitmap && r < max_rsi && good >= sp.lo && good < sp.hi && good < end_bit) {
                TrRec rec;
                uint32_t wv, ix;
                if (next_at) {
                    rec = sp.rec[next_at - 1u];
                    wv = (uint32_t)((good - sp.lo) / sp.core);
                    ix = next_at - 1u;
                } else if (!tw_lookup(sp, good, rec, wv, ix)) {
                    break;
                } else {
                    ix += sp.nbase[wv];
                }
                // (ix: the node's record index; inside the wide table its number among the nodes of the chunk's first windows)
                const bool at_front = sp.wide && (wv % sp.wpc) < sp.wfirst;
                if (at_front) ix -= sp.nbase[wv - wv % sp.wpc];
                n_lookups++;
                if (at_front && ix < sp.wcap) {
                    const uint4 wd = sp.wide[(uint64_t)(wv / sp.wpc) * sp.wcap + ix];
                    if (wd.w && r + wd.z <= max_rsi) {
                        if (lane == 0) centry[wv / sp.wpc] = ChunkEntry{good, r, 1u, 0u};
                        good = (uint64_t)wd.x | ((uint64_t)wd.y << 32);
                        r += wd.z;
                        hopped = true;
                        next_at = 0;
                        continue;
                    }
                }
                const uint32_t k1 = tr_rec_k(rec.x), b1 = tr_rec_bits(rec.x);
                if (k1 == 1u) {
                    if (lane == 0) rsi_off[r] = tr_rsi_start(c, good);
                    next_at = rec.y;
                } else if (k1 > 1u && r + k1 <= max_rsi && nh < hop_cap) {
                    if (lane == 0) hops[nh] = IdxHop{good, r, k1, 0u};      // (k_texpand writes the starts inside)
                    nh++;
                    next_at = 0;
                } else {
                    break;
                }
                good += b1;
                r += k1;
                hopped = true;
            }
            if (r >= max_rsi) {
                if ((c.flags & F_PAD_RSI) && (good & 7u) && good > start_bit) good = (good + 7u) & ~7ull;   // (as above)
                break;
            }
            if (carry && !last && good >= (s2.bitmap ? s2.hi : sp.hi)) {       // the next span continues from here
                if (lane == 0) {
                    carry->good = good;
                    carry->r = r;
                    carry->active = 1;
                    carry->n_hops = nh;
                    carry->n_serial = n_serial;
                    carry->n_lookups = n_lookups;
                }
                return;
            }
            if ((c.flags & F_PAD_RSI) && (good & 7u)) {      // reference decode.c:407-408
                good = (good + 7u) & ~7ull;
                hopped = true;
            }
            if (stop_near && end_bit - good < stop_near && (hopped || r != 0)) break;   // (status OK, b = 0: ends on an RSI start)
            if (serial_cap && (uint64_t)n_serial * c.rsi > serial_cap) {
                if (lane == 0 && carry) {
                    carry->active = 0;
                    carry->n_hops = 0;
                }
                return;
            }
            if (lane == 0) rsi_off[r] = good;
            cur_start = good;
            n_serial++;
#ifdef AEC_TUNING
            if (lane == 0 && g_dbg_serial && n_serial <= 400u) printf("serial: RSI %llu at bit %llu\n", (unsigned long long)r, (unsigned long long)good);
#endif
        }
        // keep the whole next CDS (and the readers' look-ahead) inside the LDS window
        if ((good >> 5) + (coop ? 66u : maxw + 2u) > base + kIdxWindowWords) {
            __syncthreads();
            refill(good >> 5);
            br.init(LdsWindowFetch{win, base}, end_bit, good);
            if (coop) load_regs(good >> 5);
        } else if (hopped) {
            br.init(LdsWindowFetch{win, base}, end_bit, good);
        }
        // ---- the coded data sets that begin in the next 64 bits, parsed by the 64 lanes at once; the chain through them
        // a lane read each (the cooperative parse below: ~1 us per coded data set, cross-lane round trips one behind
        // the other; this: ~0.5 us per 64 bits + 0.05 us per coded data set).  Ends on the RSI's last block (the RSI
        // start behind it is the outer loop's: tables, padding, the caller's bound), on a coded data set the piece
        // tables do not resolve, or on a zero run that does not fit (both: the parse below gives the verdict).
        if (stretch) {
            uint32_t rel0 = (uint32_t)(good - base * 32u);
            if (piece_base != base || rel0 < pc0 || rel0 + 64u > pc0 + kIdxPiece) {
                // (a piece whose look-ahead leaves the window while the stream goes on: a fresh window first)
                if ((rel0 & ~31u) + pwords * 32u > kIdxWindowWords * 32u && base + kIdxWindowWords < nwords && rel0 >= 2048u) {
                    __syncthreads();
                    refill(good >> 5);
                    br.init(LdsWindowFetch{win, base}, end_bit, good);
                    if (coop) load_regs(good >> 5);
                    rel0 = (uint32_t)(good - base * 32u);
                }
                build_piece(rel0);
            }
            const uint32_t E0 = fentry(rel0 + lane, 0u);
            const uint32_t E1 = pp ? fentry(rel0 + lane, 1u) : E0;
            uint32_t l = 0;
            bool moved = false;
            while (l < 64u) {
                const uint32_t rf = (pp && b == 0) ? 1u : 0u;
                const uint32_t e = rdlane(rf ? E1 : E0, l);
                const uint32_t len = e & 0xFFFu;
                if (!(e >> 12)) break;
                uint32_t nb1 = 1;
                if (e & kNxtZero) {
                    nb1 = spec_run_blocks(c, len - c.id_len - 1u - rf * c.bps, b);
                    if (!nb1) break;
                }
                l += len;
                good += len;
                b += nb1;
                moved = true;
                if (b >= c.rsi) {
                    b = 0;
                    r++;
                    break;
                }
            }
            if (moved) {
                hopped_far = true;                         // (the readers below start from `good` again)
                continue;
            }
        }
        const uint32_t ref = (pp && b == 0) ? 1u : 0u;
        uint32_t nblk = 1;
        bool done = false;
        if (hopped_far) {
            br.init(LdsWindowFetch{win, base}, end_bit, good);
            hopped_far = false;
        }
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
            uint32_t st = skip_cds(br, c, ref, b, nblk);
            // (the window guarantees an encoder's longest coded data set; a foreign encoder's may be longer -- 12 800 bits
            // in the tests -- and run out of the window, which reads as zeros up to the end of the stream: "input ended".
            // From a window that begins at this coded data set it is read again; the window holds 131 kbit.)
            if (st == DEC_NEED_INPUT && (good >> 5) > base + 4u && base + kIdxWindowWords < nwords) {
                __syncthreads();
                refill(good >> 5);
                if (coop) load_regs(good >> 5);
                br.init(LdsWindowFetch{win, base}, end_bit, good);
                st = skip_cds(br, c, ref, b, nblk);
            }
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
        carry->n_serial = n_serial;
        carry->n_lookups = n_lookups;
    }
    if (lane == 0 && chunk_off && batch_nhops) batch_nhops[blockIdx.x] = nh;
    if (lane == 0 && delivered) *delivered = 1u;
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

}  // namespace

namespace {


// ---- sparse path: geometry and workspace ------------------------------------------------------------
struct Sparse2Plan {
    bool ok;
    Spec2Geom g;
    size_t lds;
    // the dense fallback (k_spec4 in windows of 32 kbit with room for a candidate per four bits): geometry, windows per
    // span, offsets of its tables behind the ordinary ones
    bool dense;
    Spec2Geom dg;
    size_t dlds;
    uint32_t dnwin_max;
    size_t od_bitmap, od_pre, od_rec, od_cpos, od_ccnt, od_list;
    uint32_t nwin_max;        // windows per super-chunk (one k_spec2 launch)
    uint32_t wpc;             // windows per chunk of the wide walker
    uint32_t nchunk_max;
    // byte offsets inside the workspace (behind the 64-byte carry record)
    size_t o_bitmap, o_pre, o_rec, o_cpos, o_ccnt, o_wide, o_centry, o_hops, o_rhops, o_nhops, o_blist, bytes;
};

constexpr uint32_t kS2WindowBits = 65536;      // lead-in + core + look-ahead (16-bit positions in LDS)
constexpr uint32_t kS4WindowBits = 49152;      // k_spec4: + 2 bytes per bit for the table of coded data set lengths
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
    const bool off = tune("AEC_IDX_NO_SPARSE", 0) != 0;                  // A/B switch for measurements
    if (off || (c.flags & F_PAD_RSI) || !rsi_bits_hint || total_bits < 16384) return p;
    const uint64_t samples = (uint64_t)c.rsi * c.bs;
    // more than 4.5 bits per sample -- unless the coded data sets are still short (small blocks): what the lead-in must
    // cover is the distance a chain needs to fall onto the true one, ~2 cds^2 bits, whatever the bits per sample
    // (round 5: 8-sample blocks at up to ~9 bits per sample came over the trunk and paid its fixed phases, 17 - 50 ms
    // for 1 - 4 MiB: tests/fuzz_index_gpu.py --time)
    const uint64_t cds_max = tune("AEC_S2_CDS_MAX", 80);
    // (with the preprocessor only: raw samples repeat the same bits at the same place, a chain that is off the true one
    // does not find it again, and every RSI fell to the serial walker -- 16 MiB of 8-bit data, ratio 1.19: 1.4 s)
    if (rsi_bits_hint * 2 > samples * 9 && (rsi_bits_hint > (uint64_t)c.rsi * cds_max || !(c.flags & F_PREPROCESS))) return p;
    uint64_t look = (2 * rsi_bits_hint + 1024 + 31) & ~31ull;
    if (look < 4096) look = 4096;
    uint32_t wbits = tune("AEC_S2_WINDOW", kS2WindowBits);
    if (wbits < 16384 || wbits > kS2WindowBits) wbits = kS2WindowBits;
    if (look + kS2Lead + 8192 > wbits) wbits = kS2WindowBits;            // long RSIs: the largest window
    if (look > kS2WindowBits - kS2Lead - 16384) return p;
    // k_spec4: the table of every position's coded data set wants 2 bytes per bit of the window, so the window is 48
    // kbit; candidates have room for one per 12 bits (coded data sets of 16 bits and more on average)
    // Measured (profiles/r05): per window it is 20 - 40 % cheaper than k_spec2, but the windows are a quarter smaller --
    // a wash on large streams (1 GiB of config 2: 21.3 against 20.1 ms), a gain where the input is a few windows that do
    // not fill the chip anyway (a 64 KiB chunk: 35 us of 250 per call).  So: small inputs only (up to 4 MB of stream).
    const bool small = total_bits <= (1ull << 25);
    const bool v4 = tune("AEC_S2_V4", small ? 1u : 0u) != 0 && look + kS2Lead + 8192 <= kS4WindowBits &&
                    rsi_bits_hint >= (uint64_t)c.rsi * 16u && !tune_set("AEC_S2_WINDOW");
    if (v4) wbits = kS4WindowBits;
    uint64_t core = (wbits - kS2Lead - look) & ~1023ull;
    // small inputs: about one window per CU
    const uint64_t want = ((total_bits / 256 + 1023) & ~1023ull);
    if (core > want) core = want < 8192 ? 8192 : want;
    p.g.lead = kS2Lead;
    p.g.core = (uint32_t)core;
    p.g.look = (uint32_t)look;
    p.g.stride = tune("AEC_S2_STRIDE", 64u);
    p.g.burn = tune("AEC_S2_BURN", 24u);
    p.g.fast = tune("AEC_S2_FAST", 1u);
    p.g.refill = tune("AEC_S2_REFILL", 16u);
    // (headers of one or two bits -- AEC_RESTRICTED with at most 4 bits per sample -- are no pattern to go by: eight
    // in a row happen by chance all the time there)
    p.g.uncrun = tune("AEC_S2_UNCRUN", c.id_len >= 3u ? 1u : 0u);
    const uint32_t W = p.g.lead + p.g.core + p.g.look, nw = W / 32;
    const uint32_t capdiv = tune("AEC_S2_CAPDIV", v4 ? 12u : 8u);
    p.g.cap_lds = (W / (capdiv ? capdiv : 8u) + 63) & ~63u;
    p.g.cap_core = (p.g.core / 8 + 63) & ~63u;
    p.g.v4 = v4 ? 1u : 0u;
    p.lds = (size_t)(nw + 4) * 4 + (size_t)nw * 4 + (size_t)(nw + 2) * 2 * 3 + (size_t)p.g.cap_lds * 2 * 7 + 64;
    if (v4) {
        if (p.g.cap_core > p.g.cap_lds) p.g.cap_core = p.g.cap_lds;
        const size_t tab = (size_t)W * 2 > (size_t)p.g.cap_lds * 2 * 3 ? (size_t)W * 2 : (size_t)p.g.cap_lds * 2 * 3;
        p.lds = (size_t)(nw + 4) * 4 + (size_t)nw * 4 + (size_t)(nw + 2) * 2 * 3 + (size_t)p.g.cap_lds * 2 * 4 + tab + 64;
    }
    if (p.lds > 156 * 1024) return p;
    const uint64_t nwin_total = (total_bits + core + core - 1) / core;    // (+ one: the range starts on a core boundary)
    p.nwin_max = (uint32_t)(nwin_total < kS2SuperWindows ? nwin_total : kS2SuperWindows);
    // (windows per chunk: a lane of k_wide / k_rewalk follows the records through ONE chunk, ~1 us per window, and the
    // walker takes a lookup per chunk, ~2 us -- chunks of 256 windows made the three of them 0.6 ms of latency per span,
    // 17 % of the index time of a 64 MiB stream; measured 16 .. 256: 32 is best from 64 MiB to 1 GiB)
    p.wpc = p.nwin_max >= 512 ? 32u : (p.nwin_max >= 64 ? 16u : p.nwin_max);
    p.wpc = tune("AEC_S2_WPC", p.wpc);
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
    p.o_blist = o;  o = up(o + (size_t)kS2BridgeCap * 4);
    // Dense fallback.  A stream whose coded data sets repeat -- a ramp, a gradient: every block the same residuals -- is
    // periodic, chains from different places fall into different cycles and never meet, and the candidates of all of
    // them exceed a window's room: every window gave up and ONE wavefront walked the stream (64 MiB of a 16-bit ramp:
    // 2 s, the reference 0.1 s).  Windows half the size with four times the room per bit resolve those; they are built
    // only where an ordinary window gave up, and the walker takes their records RSI by RSI.
    p.dense = false;
    if (tune("AEC_S2_DENSE", 1) && look + kS2Lead + 4096 <= 32768) {
        p.dg = p.g;
        p.dg.lead = kS2Lead;
        p.dg.look = (uint32_t)look;
        p.dg.core = (32768u - kS2Lead - (uint32_t)look) & ~1023u;
        if (p.dg.core > p.g.core) p.dg.core = p.g.core;
        p.dg.cap_lds = 8192;
        p.dg.cap_core = p.dg.core / 4;
        p.dg.v4 = 1;
        const uint32_t dW = p.dg.lead + p.dg.core + p.dg.look, dnw = dW / 32;
        const size_t tab = (size_t)dW * 2 > (size_t)p.dg.cap_lds * 2 * 3 ? (size_t)dW * 2 : (size_t)p.dg.cap_lds * 2 * 3;
        p.dlds = (size_t)(dnw + 4) * 4 + (size_t)dnw * 4 + (size_t)(dnw + 2) * 2 * 3 + (size_t)p.dg.cap_lds * 2 * 4 + tab + 64;
        const uint64_t dn = ((uint64_t)p.nwin_max * p.g.core + p.dg.core - 1) / p.dg.core + 1;
        if (p.dlds <= 156 * 1024 && dn < (1u << 24)) {
            p.dnwin_max = (uint32_t)dn;
            const size_t dwords = (size_t)p.dnwin_max * (p.dg.core / 32);
            p.od_bitmap = o; o = up(o + dwords * 4);
            p.od_pre = o;    o = up(o + dwords * 2);
            p.od_rec = o;    o = up(o + (size_t)p.dnwin_max * p.dg.cap_core * sizeof(uint2));
            p.od_cpos = o;   o = up(o + (size_t)p.dnwin_max * p.dg.cap_core * 2);
            p.od_ccnt = o;   o = up(o + (size_t)p.dnwin_max * 4);
            p.od_list = o;   o = up(o + (size_t)p.dnwin_max * 4);
            p.dense = true;
        }
    }
    p.bytes = o;
    p.ok = true;
    return p;
}

// AEC_S2_PROF=1: phase stamps of k_spec2 (diagnostics; printed by the host at exit of the first launch)
unsigned long long *spec2_prof_buffer(uint32_t nwin, bool reset = true)
{
    static const bool on = tune("AEC_S2_PROF", 0) != 0;
    static unsigned long long *buf = nullptr;
    if (!on) return nullptr;
    if (!buf) (void)hipMalloc(reinterpret_cast<void **>(&buf), (size_t)kS2SuperWindows * 16 * sizeof(unsigned long long));
    if (reset) (void)hipMemset(buf, 0, (size_t)kS2SuperWindows * 16 * sizeof(unsigned long long));
    (void)nwin;
    return buf;
}

void spec2_prof_report(uint32_t nwin, hipStream_t st, bool v4 = false)
{
    unsigned long long *buf = spec2_prof_buffer(nwin, false);
    if (!buf) return;
    static int reports = 0;
    if (reports++ >= 2) return;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nwin * 16);
    (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
    if (v4) {                                   // k_spec4: stamps 0 .. 7 in the order of the phases
        double e[7] = {0, 0, 0, 0, 0, 0, 0};
        uint32_t m = 0;
        for (uint32_t w = 0; w + 1 < nwin; w++) {
            if (!h[(size_t)w * 16 + 7]) continue;
            for (int k = 0; k < 7; k++) e[k] += (double)(h[(size_t)w * 16 + k + 1] - h[(size_t)w * 16 + k]);
            m++;
        }
        if (!m) m = 1;
        double f1 = 0, f2 = 0;
        for (uint32_t w = 0; w + 1 < nwin; w++) {
            if (!h[(size_t)w * 16 + 7]) continue;
            f1 += (double)(h[(size_t)w * 16 + 8] - h[(size_t)w * 16 + 3]);
            f2 += (double)(h[(size_t)w * 16 + 9] - h[(size_t)w * 16 + 8]);
        }
        fprintf(stderr, "k_spec4 phases (shader-clock ticks per window, %u windows): load+rank %.0f | table %.0f | chains %.0f | "
                "candidates+first+catch-up %.0f (candidates %.0f, first coded data sets %.0f) | hop tables %.0f | table steps %.0f | chain+write %.0f\n", m, e[0] / m,
                e[1] / m, e[2] / m, e[3] / m, f1 / m, f2 / m, e[4] / m, e[5] / m, e[6] / m);
        return;
    }
    double d[6] = {0, 0, 0, 0, 0, 0};
    uint32_t n = 0;
    for (uint32_t w = 0; w + 1 < nwin; w++) {
        if (!h[(size_t)w * 16 + 6]) continue;
        for (int k = 0; k < 6; k++) d[k] += (double)(h[(size_t)w * 16 + k + 1] - h[(size_t)w * 16 + k]);
        n++;
    }
    double pa = 0;
    for (uint32_t w = 0; w + 1 < nwin; w++)
        if (h[(size_t)w * 16 + 6]) pa += (double)(h[(size_t)w * 16 + 7] - h[(size_t)w * 16 + 4]);
    fprintf(stderr, "k_spec2 phases (shader-clock ticks per window, %u windows): load+rank %.0f | chains %.0f | "
            "prefix+cand nxt %.0f | hop4+hop16 %.0f | units %.0f (parses %.0f) | chain+write %.0f\n", n, d[0] / n,
            d[1] / n, d[2] / n, d[3] / n, d[4] / n, pa / n, d[5] / n);
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
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_spec4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                156 * 1024) != hipSuccess)
            (void)hipGetLastError();

    });
}

// Index pass over the sparse tables, super-chunk by super-chunk: speculation (all CUs), wide walker
// (every candidate of every chunk's first window), the walk (one wavefront: one lookup per chunk where
// the wide table resolves it, per window or per coded data set where not), then the true chain through
// the skipped chunks and the RSI starts inside all hops.
// spans from which on the pipeline is worth a second set of tables (a quarter of a GB; below, e.g. for the
// 256 MiB calls of the streaming ABI, one set is kept and the spans run behind one another)
constexpr uint64_t kS2PipeSpans = 4;

// A side stream with its events, for the span pipeline below.  Created on first use, kept for the life of the process
// (a few per device: concurrent callers each take their own), never destroyed while work may be pending.
struct SideStream {
    int device = -1;
    hipStream_t st = nullptr;
    hipEvent_t start = nullptr, tab[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
};
std::mutex g_side_mu;
std::vector<SideStream> *g_side_free = nullptr;

bool side_take(SideStream *out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        if (g_side_free)
            for (size_t i = 0; i < g_side_free->size(); i++)
                if ((*g_side_free)[i].device == dev) {
                    *out = (*g_side_free)[i];
                    g_side_free->erase(g_side_free->begin() + (ptrdiff_t)i);
                    return true;
                }
    }
    SideStream s;
    s.device = dev;
    bool ok = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) == hipSuccess;
    for (hipEvent_t *e : {&s.start, &s.tab[0], &s.tab[1], &s.done[0], &s.done[1]})
        ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        return false;                                   // (what was created is lost: a one-time leak of a failed start)
    }
    *out = s;
    return true;
}

void side_give(const SideStream &s)
{
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (!g_side_free) g_side_free = new (std::nothrow) std::vector<SideStream>();
    if (g_side_free) g_side_free->push_back(s);
}

// Spans of at most nwin_max windows.  With room for two sets of tables the spans are pipelined over two streams: the
// table kernel of span s + 1 (k_spec2: all CUs) runs on a side stream beside the walkers of span s (k_wide, k_index,
// k_rewalk, k_expand2: a few hundred wavefronts down to one, 16 % of the time of a span when they ran behind one another), ordered by
// events; `st` has waited for everything the side stream did when the last walker is enqueued.
void launch_index_sparse(const Cfg &c, const Sparse2Plan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                         uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                         uint8_t *base, size_t ws_bytes, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot,
                         uint64_t stop_near, const uint32_t *skip_if = nullptr, uint32_t serial_cap = 0,
                         uint32_t *delivered = nullptr)
{
    allow_big_lds2();
    Spec2Geom geom = p.g, dgeom = p.dg;
    geom.skip_if = dgeom.skip_if = skip_if;
    IdxCarry *carry = reinterpret_cast<IdxCarry *>(base);
    const uint64_t lo0 = start_bit / p.g.core * p.g.core;
    const uint64_t span = (uint64_t)p.nwin_max * p.g.core;
    const uint64_t nspans = (end_bit - lo0 + span - 1) / span;
    SideStream side;
    bool piped = nspans >= kS2PipeSpans && ws_bytes >= 2 * p.bytes && !spec2_prof_buffer(0, false);
#ifdef AEC_TUNING
    if (tune_set("AEC_S2_VERIFY") || tune("AEC_S2_PIPE", 1) == 0) piped = false;
#endif
    piped = piped && side_take(&side);
    if (piped) {
        (void)hipEventRecord(side.start, st);                       // the input is whatever `st` has produced so far
        (void)hipStreamWaitEvent(side.st, side.start, 0);
    }
    uint64_t si = 0;
    for (uint64_t lo = lo0; lo < end_bit; lo += span, si++) {
        const uint64_t bits = end_bit - lo < span ? end_bit - lo : span;
        const uint32_t nwin = (uint32_t)((bits + p.g.core - 1) / p.g.core);
        const uint32_t nchunks = (nwin + p.wpc - 1) / p.wpc;
        const bool first = lo == lo0, last = lo + span >= end_bit;
        const uint32_t set = piped ? (uint32_t)(si & 1u) : 0u;
        uint8_t *tb = base + (size_t)set * p.bytes;                  // (the carry record lives in set 0's header)
        hipStream_t ts = piped ? side.st : st;                       // stream of the table kernels
        SparseTables t;
        t.bitmap = reinterpret_cast<const uint32_t *>(tb + p.o_bitmap);
        t.pre = reinterpret_cast<const uint16_t *>(tb + p.o_pre);
        t.rec = reinterpret_cast<const uint2 *>(tb + p.o_rec);
        t.cpos = reinterpret_cast<const uint16_t *>(tb + p.o_cpos);
        t.ccnt = reinterpret_cast<const uint32_t *>(tb + p.o_ccnt);
        t.lo = lo;
        t.hi = lo + (uint64_t)nwin * p.g.core;
        t.core = p.g.core;
        t.cap = p.g.cap_core;
        t.wide = reinterpret_cast<const uint4 *>(tb + p.o_wide);
        t.wpc = p.wpc;
        t.skip_if = skip_if;
        ChunkEntry *centry = reinterpret_cast<ChunkEntry *>(tb + p.o_centry);
        IdxHop *hops = reinterpret_cast<IdxHop *>(tb + p.o_hops);
        IdxHop *rhops = reinterpret_cast<IdxHop *>(tb + p.o_rhops);
        uint32_t *nhops = reinterpret_cast<uint32_t *>(tb + p.o_nhops);
        const uint32_t hop_cap = 2 * nwin + 8;
        if (piped && si >= 2) (void)hipStreamWaitEvent(ts, side.done[set], 0);      // the walkers of span si - 2 are through
        uint32_t *blist = reinterpret_cast<uint32_t *>(tb + p.o_blist), *blist_cnt = reinterpret_cast<uint32_t *>(tb + 56);
        (void)hipMemsetAsync(blist_cnt, 0, 8, ts);                  // (+ the count of the dense windows' list at offset 60)
        if (p.g.v4)
            hipLaunchKernelGGL(k_spec4, dim3(nwin), dim3(1024), p.lds, ts, c, words, nwords, end_bit, lo, start_bit, geom,
                               const_cast<uint32_t *>(t.bitmap), const_cast<uint16_t *>(t.pre), const_cast<uint2 *>(t.rec),
                               const_cast<uint16_t *>(t.cpos), const_cast<uint32_t *>(t.ccnt), spec2_prof_buffer(nwin),
                               (const uint64_t *)nullptr, 0u, blist, blist_cnt);
        else
            hipLaunchKernelGGL(k_spec2, dim3(nwin), dim3(1024), p.lds, ts, c, words, nwords, end_bit, lo, start_bit, geom,
                               const_cast<uint32_t *>(t.bitmap), const_cast<uint16_t *>(t.pre), const_cast<uint2 *>(t.rec),
                               const_cast<uint16_t *>(t.cpos), const_cast<uint32_t *>(t.ccnt), spec2_prof_buffer(nwin),
                               (const uint64_t *)nullptr, 0u, blist, blist_cnt);
        spec2_prof_report(nwin, ts, p.g.v4 != 0);
        SparseTables td{};
        if (p.dense) {
            const uint32_t dnwin = (uint32_t)(((uint64_t)nwin * p.g.core + p.dg.core - 1) / p.dg.core);
            td.bitmap = reinterpret_cast<const uint32_t *>(tb + p.od_bitmap);
            td.pre = reinterpret_cast<const uint16_t *>(tb + p.od_pre);
            td.rec = reinterpret_cast<const uint2 *>(tb + p.od_rec);
            td.cpos = reinterpret_cast<const uint16_t *>(tb + p.od_cpos);
            td.ccnt = reinterpret_cast<const uint32_t *>(tb + p.od_ccnt);
            td.lo = lo;
            td.hi = lo + (uint64_t)dnwin * p.dg.core < t.hi ? lo + (uint64_t)dnwin * p.dg.core : t.hi;
            td.core = p.dg.core;
            td.cap = p.dg.cap_core;
            td.wide = nullptr;
            td.wpc = 1;
            td.skip_if = skip_if;
            uint32_t *dlist = reinterpret_cast<uint32_t *>(tb + p.od_list), *dcount = reinterpret_cast<uint32_t *>(tb + 60);
            if (dnwin <= 1024u)
                hipLaunchKernelGGL(k_spec4, dim3(dnwin), dim3(1024), p.dlds, ts, c, words, nwords, end_bit, lo, start_bit, dgeom,
                                   const_cast<uint32_t *>(td.bitmap), const_cast<uint16_t *>(td.pre), const_cast<uint2 *>(td.rec),
                                   const_cast<uint16_t *>(td.cpos), const_cast<uint32_t *>(td.ccnt), (unsigned long long *)nullptr,
                                   (const uint64_t *)nullptr, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                   (const uint32_t *)nullptr, t.ccnt, p.g.core, nwin);
            else {
            hipLaunchKernelGGL(k_dense_pick, dim3((dnwin + 255) / 256), dim3(256), 0, ts, t.ccnt, p.g.core, nwin, p.dg.core, dnwin,
                               const_cast<uint32_t *>(td.ccnt), dlist, dcount, skip_if);
            hipLaunchKernelGGL(k_spec4, dim3(dnwin < 512u ? dnwin : 512u), dim3(1024), p.dlds, ts, c, words, nwords, end_bit, lo,
                               start_bit, dgeom, const_cast<uint32_t *>(td.bitmap), const_cast<uint16_t *>(td.pre),
                               const_cast<uint2 *>(td.rec), const_cast<uint16_t *>(td.cpos), const_cast<uint32_t *>(td.ccnt),
                               (unsigned long long *)nullptr, (const uint64_t *)nullptr, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr,
                               (const uint32_t *)dlist, (const uint32_t *)dcount);
            }
        }
#ifdef AEC_TUNING
        if (tune_set("AEC_S2_VERIFY")) {
            static uint32_t *d_bad = nullptr;
            if (!d_bad) (void)hipMalloc(reinterpret_cast<void **>(&d_bad), 8);
            (void)hipMemsetAsync(d_bad, 0, 8, st);
            hipLaunchKernelGGL(k_spec_verify, dim3(nwin), dim3(256), 0, st, c, TrStream{words, nwords, end_bit}, t, nwin, d_bad);
            uint32_t h[2] = {0, 0};
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, d_bad, 8, hipMemcpyDeviceToHost);
            uint32_t nb = 0;
            (void)hipMemcpy(&nb, blist_cnt, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "verify: %u windows, lookup mismatches %u, wrong records %u; hypotheses for k_bridge: %u\n", nwin, h[0],
                    h[1], nb);
        }
#endif
        if (piped) {
            (void)hipEventRecord(side.tab[set], ts);
            (void)hipStreamWaitEvent(st, side.tab[set], 0);
        }
        // (a few windows only -- a chunk of an HDF5 dataset: the walker takes a lookup per window itself, which costs
        // what the chunk-level chase alone would, and five launches less)
        const bool flat = nwin <= kS2FlatWindows;
        if (flat) t.wide = nullptr;
        // (the chunk-level chase has a few hundred wavefronts: with the walkers, beside the next span's k_spec2)
        if (!flat) {
            (void)hipMemsetAsync(tb + p.o_wide, 0, (size_t)nchunks * p.g.cap_core * sizeof(uint4), st);
            (void)hipMemsetAsync(centry, 0, (size_t)nchunks * sizeof(ChunkEntry), st);
        }
        if (tune("AEC_S2_BRIDGE", 1))
            hipLaunchKernelGGL(k_bridge, dim3(1024), dim3(64), 0, st, c, words, nwords, end_bit, t,
                               const_cast<uint2 *>(t.rec), blist, blist_cnt, carry, first ? 1u : 0u,
                               (uint32_t)tune("AEC_S2_BRIDGE_AFTER", kS2BridgeAfter));
        if (!flat)
            hipLaunchKernelGGL(k_wide, dim3((p.g.cap_core + 255) / 256, nchunks), dim3(256), 0, st, t, nwin, end_bit,
                               const_cast<uint4 *>(t.wide));
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi,
                           d_res, (const uint64_t *)nullptr, hops, hop_cap, carry, first ? 1u : 0u, last ? 1u : 0u,
                           start_block, rsi_start, tail_slot, TwTables{}, centry, t, (uint32_t *)nullptr, last ? stop_near : 0ull,
                           skip_if, td, (first && last) ? serial_cap : 0u, (first && last) ? delivered : (uint32_t *)nullptr);
        if (!flat)
            hipLaunchKernelGGL(k_rewalk, dim3((nchunks + 63) / 64), dim3(64), 0, st, t, nwin, nchunks, end_bit, centry, rhops,
                               nhops, d_rsi_off);
        hipLaunchKernelGGL(k_expand2, dim3((hop_cap + 255) / 256), dim3(256), 0, st, t, carry, hops,
                           (const uint32_t *)nullptr, 0u, 0u, d_rsi_off);
        if (!flat)
            hipLaunchKernelGGL(k_expand2, dim3((nchunks * p.wpc * 2 + 255) / 256), dim3(256), 0, st, t, carry, rhops, nhops,
                               nchunks, p.wpc * 2, d_rsi_off);
        if (piped) (void)hipEventRecord(side.done[set], st);
#ifdef AEC_TUNING
        if (tune_set("AEC_IDX_STATS")) {                   // (diagnostics: synchronises)
            (void)hipStreamSynchronize(st);
            const int dbg = tune("AEC_IDX_STATS", 0) >= 2 ? 1 : 0;
            const hipError_t de = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_serial), &dbg, sizeof(dbg));
            if (de != hipSuccess) fprintf(stderr, "g_dbg_serial: %s\n", hipGetErrorString(de));
            std::vector<uint32_t> cc(nwin);
            (void)hipMemcpy(cc.data(), t.ccnt, (size_t)nwin * 4, hipMemcpyDeviceToHost);
            uint32_t gave_up = 0, most = 0, nb = 0;
            uint64_t sum = 0;
            for (uint32_t v : cc) {
                gave_up += v == 0;
                most = v > most ? v : most;
                sum += v;
            }
            IdxCarry h{};
            (void)hipMemcpy(&h, carry, sizeof(h), hipMemcpyDeviceToHost);
            (void)hipMemcpy(&nb, blist_cnt, 4, hipMemcpyDeviceToHost);
            {
                const uint32_t m = nb < kS2BridgeCap ? nb : kS2BridgeCap;
                std::vector<uint32_t> bl(m);
                if (m) (void)hipMemcpy(bl.data(), blist, (size_t)m * 4, hipMemcpyDeviceToHost);
                uint32_t hist[32] = {0};
                for (uint32_t v : bl) hist[(v >> 26) & 31u]++;
                fprintf(stderr, "  headers of the listed hypotheses:");
                for (uint32_t q = 0; q < (1u << c.id_len); q++) fprintf(stderr, " %u:%u", q, hist[q]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "window tables, span %llu: %u windows (%u gave up), candidates per window %.0f on average, %u at "
                    "most (room for %u) | so far RSIs %llu, walked serially %u, table lookups %u | hypotheses listed for "
                    "k_bridge %u\n", (unsigned long long)si, nwin, gave_up, nwin ? (double)sum / nwin : 0.0, most,
                    p.g.cap_core, (unsigned long long)h.r, h.n_serial, h.n_lookups, nb);
        }
#endif
    }
    if (piped) side_give(side);
}


// ---- geometry and workspace of the trunk index -------------------------------------------------------
struct TrunkPlan {
    bool ok;
    uint32_t L, lead, rw, passes, budget, kmax, wpw, wpc, wcap, wfirst;
    uint32_t staged, margin;  // hypothesis walks: byte table of coded data set lengths or not, margin behind a group (bits)
    size_t lds;               // ... and the LDS of a workgroup
    // coalescing hypothesis walks (aec_trunk.h section 2b): windows per group, margin, bits per mark cell (log2),
    // parses before a walk is handed on, walks per group, LDS, list capacities
    uint32_t co, co_wpg, co_margin, co_shift, co_tmax, co_cap, co_qcap, co_pcap, co_over, co_park;
    size_t co_lds, o_coq, o_cop;
    uint32_t pcap;            // pool of RSI ends inside records
    uint32_t nwin_max;        // windows per span (launch set): core + look-ahead
    uint32_t nlook;           // look-ahead windows behind the core of a span that does not reach the end
    uint32_t ncap;            // node records per span
    // byte offsets inside the workspace (behind the 64-byte carry record)
    size_t o_bitmap, o_pre, o_nbase, o_cpos, o_bp, o_ccnt, o_nblk, o_nros, o_entry, o_exit0, o_exit1, o_gbase, o_seam,
        o_ros, o_rec, o_park, o_pool, o_wide, o_centry, o_hops, bytes;
};

// workspace asked for at most (larger inputs take several spans): 768 MiB, for large streams up to six times the
// stream and 4 GiB -- several kernels of a span take as long as the longest serial chain they hold (a trunk region,
// a walk that was handed on, the walker's hops), whatever the span's size, so fewer and larger spans are cheaper
constexpr size_t kTrWsWanted = 768u << 20, kTrWsMost = (size_t)4 << 30;


size_t trunk_bytes(TrunkPlan &p, uint32_t nwin)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t words = (size_t)nwin * (p.L / 32);
    const uint32_t nchunk = (nwin + p.wpc - 1) / p.wpc;
    size_t o = 128;                     // (a header: carry record, pool counter, list counters, statistics)
    p.o_bitmap = o; o = up(o + words * 4);
    p.o_pre = o;    o = up(o + words * 2);
    p.o_nbase = o;  o = up(o + ((size_t)nwin + 1) * 4);
    p.o_cpos = o;   o = up(o + (size_t)p.ncap * 2);
    p.o_bp = o;     o = up(o + (size_t)p.ncap * 4);
    p.o_ccnt = o;   o = up(o + (size_t)nwin * 4);
    p.o_nblk = o;   o = up(o + (size_t)nwin * 4);
    p.o_nros = o;   o = up(o + (size_t)nwin * 4);
    p.o_entry = o;  o = up(o + (size_t)nwin * 8);
    p.o_exit0 = o;  o = up(o + (size_t)nwin * 8);
    p.o_exit1 = o;  o = up(o + (size_t)nwin * 8);
    p.o_gbase = o;  o = up(o + ((size_t)nwin + 1) * 8);
    p.o_seam = o;   o = up(o + ((size_t)nwin + 1) * 4);
    p.o_ros = o;    o = up(o + ((size_t)nwin + 1) * 4);
    p.o_rec = o;    o = up(o + (size_t)p.ncap * sizeof(TrRec));
    p.o_park = o;   o = up(o + (size_t)p.ncap * 4);
    p.o_pool = o;   o = up(o + (size_t)p.pcap * sizeof(TrPoolEntry));
    p.o_wide = o;   o = up(o + (size_t)nchunk * p.wcap * sizeof(uint4));
    p.o_centry = o; o = up(o + (size_t)nchunk * sizeof(ChunkEntry));
    p.o_hops = o;   o = up(o + ((size_t)nwin * 2 + 16) * sizeof(IdxHop));
    if (p.co) {
        p.co_qcap = p.ncap / 4u + 4096u;
        p.co_pcap = p.ncap / 8u + 4096u;
        p.o_coq = o; o = up(o + (size_t)p.co_qcap * sizeof(uint2));
        p.o_cop = o; o = up(o + (size_t)p.co_pcap * sizeof(uint2));
    }
    return o;
}

// The trunk pays off as soon as the stream is long enough to give every lane a region of its own; the
// geometry follows from the average coded data set (caller's estimate of the coded RSI / blocks per RSI):
// the self-synchronising parse needs about 2 * cds^2 bits to fall onto another chain (measured in tests/emul:
// 4.5 kbit at 45 bits per coded data set, 126 kbit at 253), which sizes burn-in and region.
// ws_bytes: what the caller can give (0 = say what is wanted).
TrunkPlan trunk_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint, size_t ws_bytes)
{
    TrunkPlan p{};
    if (total_bits < 32768) return p;
    uint64_t cds = rsi_bits_hint ? rsi_bits_hint / c.rsi : (uint64_t)(c.id_len + c.bs * c.bps) / 3;
    if (cds < 8) cds = 8;
    if (cds > 4096) cds = 4096;
    const uint64_t sync = 2 * cds * cds;
    // windows: 128 to 256 coded data sets (measured at 253 bits per coded data set: 81 ms per GiB with windows of
    // 32768 or 16384 bits, 91 with 65536, 212 with 8192), fewer bits for small inputs so that the lanes still fill
    // the chip
    // (never below 32768 bits, whatever the size of the input: smaller windows for small inputs -- and for short coded
    // data sets -- were the rule of round 3, and they leave most RSIs to the serial walker on many shapes: with the
    // sample shape (720 bits per coded data set) windows of 8192 bits made 20 MiB take 83 ms where 32768 take 8; 16 MiB
    // of 16-bit data at 8 bits per sample (rsi 128) 134 ms with 8192, 64 with 16384, 4.2 with 32768; 12-bit, rsi 64:
    // 150 / 150 / 4.7 ms.  Found by sweeping sizes and shapes the benches do not have: tests/fuzz_index_gpu.py --time.)
    uint32_t L = 32768;
    if (total_bits >= (1ull << 28) && 128 * cds > 32768) L = 65536;      // (sample shape: 32768 is better below ~100 MiB)
    p.L = tune("AEC_TR_L", L);
    uint64_t lead = 4 * sync;
    if (lead < 4096) lead = 4096;
    if (lead > (1u << 19)) lead = 1u << 19;            // (measured: repair passes are cheaper than longer burn-ins)
    p.lead = tune("AEC_TR_LEAD", (uint32_t)lead);
    uint64_t rw = (4 * sync + p.L - 1) / p.L;
    if (rw < 1) rw = 1;
    if (rw > 8) rw = 8;
    p.rw = tune("AEC_TR_RW", (uint32_t)rw);
    // (a run of k regions whose burn-in did not find the chain takes k repair passes, and a seam that stays costs far
    // more than a pass: the walks behind it cannot jump across -- 1 GiB of config 3 with one pass too few 36 .. 400 ms
    // instead of 24.  A pass with nothing to repair is a launch of a few microseconds, so there are two to spare.)
    // (Sixteen: on raw samples without the preprocessor -- every block uncompressed, the same bits at the same place in
    // every sample -- a chain off the true one does not find it again at all, whole runs of regions start wrong, and
    // every pass repairs one more of a run: 16 MiB 85 / 63 / 122 ms with five passes, 77 / 42 / 109 with sixteen; still
    // mostly serial, an open item.)
    p.passes = tune("AEC_TR_PASSES", 16);
    // RSIs per record: a walk meets the trunk inside one RSI with probability about 1 - exp(-rsi bits / sync);
    // enough RSIs that a true start fails once in 1e5
    {
        const double x = (double)(rsi_bits_hint ? rsi_bits_hint : (uint64_t)c.rsi * cds) / (double)sync;
        double k = x > 0.01 ? 11.5 / x : 1e9;
        if (k < 4) k = 4;
        if (k > kTrMaxK) k = kTrMaxK;
        (void)k;
        p.kmax = tune("AEC_TR_KMAX", kTrMaxK);
    }
    const uint64_t budget = (uint64_t)p.kmax * c.rsi;
    p.budget = tune("AEC_TR_BUDGET", (uint32_t)(budget < 65536 ? budget : 65536));
    // wide walker: the first windows of a chunk, whose nodes get a wide entry, hold two average RSIs; a chunk is
    // eight times that (64 windows at least)
    {
        const uint64_t rb = rsi_bits_hint ? rsi_bits_hint : (uint64_t)c.rsi * cds;
        uint64_t wf = (2 * rb + p.L - 1) / p.L + 1;
        if (wf > 512) wf = 512;
        p.wfirst = tune("AEC_TR_WFIRST", (uint32_t)wf);
        p.wpc = tune("AEC_TR_WPC", p.wfirst * 8u > 64u ? p.wfirst * 8u : 64u);
        if (p.wfirst > p.wpc) p.wfirst = p.wpc;
    }
    // The hypothesis walks run on staged stretches of the stream: group + margin.  Short coded data sets get the
    // byte table (1.25 bytes of LDS per bit, stretch 96 kbit); the others stage words and marks only (0.25 bytes
    // per bit, stretch up to 480 kbit).  The margin holds an average walk -- some 150 coded data sets where the
    // trunk resynchronises fast, the resynchronisation distance where it does not; longer walks go on in device memory.
    p.staged = tune("AEC_TR_NX", cds <= 96 ? 1u : 0u);
    if (p.staged) {
        const uint32_t stretch = tune("AEC_TR_STRETCH", 98304u);
        uint64_t margin = 512 * cds;
        if (margin > stretch * 2ull / 3) margin = stretch * 2ull / 3;
        margin = (margin + 1023) & ~1023ull;
        p.margin = tune("AEC_TR_MARGIN", (uint32_t)margin);
        uint32_t wpg = stretch > p.margin + p.L ? (stretch - p.margin) / p.L : 1u;
        if (wpg > 256u) wpg = 256u;
        p.wpw = tune("AEC_TR_WPW", wpg);
        const uint32_t bits = p.wpw * p.L + p.margin;
        p.lds = (size_t)(bits / 32 + 8) * 4 + (size_t)(bits / 32) * 4 + (size_t)(p.wpw + 1) * 8 + bits + 64;
    } else {
        // device-memory walks: a window per wavefront (measured on C3 / typical shapes: more wavefronts beat
        // longer node lists per lane -- the walks are bound by memory latency)
        p.wpw = tune("AEC_TR_WPW", 1u);
    }
    // Coalescing walks where a walk does not complete its RSI before it is back on the trunk (long RSIs: config 3) --
    // a walk then meets another walk after a dozen parses instead of the trunk after a few hundred.  Groups of up to
    // 64 kbit (about a thousand nodes at most), marks of 16 bits, 64 parses before a walk is handed on (measured in
    // tests/emul on the config-3 shape: 15 + 15 parses per node against 280; 16 + 8 with 128 parses).
    p.co = tune("AEC_TR_CO", (!p.staged && c.rsi >= 8 * cds) ? 1u : 0u);
    if (p.co) {
        // (measured on config 3, 1 GiB: what counts is that two or three workgroups fit a CU -- 96 kbit + 16 kbit of margin
        // 24.0 ms, 64 + 32 kbit 27.0, 128 + 16 kbit 28.0; walks that leave the margin are handed on, which costs little)
        uint64_t core = 1024 * cds;
        if (core > 98304) core = 98304;
        uint32_t wpg = (uint32_t)(core / p.L);
        if (wpg < 1) wpg = 1;
        p.co_wpg = tune("AEC_TR_CO_WPG", wpg);
        uint32_t margin = (uint32_t)((64u * cds + 1023u) & ~1023ull);
        margin = margin < 8192u ? 8192u : (margin > 32768u ? 32768u : margin);
        p.co_margin = tune("AEC_TR_CO_MARGIN", margin) & ~31u;
        if (p.co_margin < 2048u) p.co_margin = 2048u;
        p.co_shift = tune("AEC_TR_CO_SHIFT", 4u);
        if (p.co_shift < 2u || p.co_shift > 4u) p.co_shift = 4u;
        p.co_tmax = tune("AEC_TR_CO_TMAX", 64u);
        p.co_park = tune("AEC_TR_CO_PARK", 16u);
        p.co_cap = tune("AEC_TR_CO_CAP", 1024u);
        // walks whose count runs over the end of their RSI before they land:
        // (AEC_TR_CO_OVER=1, measurements: every such walk to the plain walk.  The walks that matter among them -- true
        // chains the trunk has not found again for a whole RSI -- are told by their counts (aec_trunk.h tr_co_rest) and
        // get the plain walk in any case; the rest are garbage walks by the thousand, each thousands of parses long.)
        p.co_over = tune("AEC_TR_CO_OVER", 0u);
        if (p.co_cap > kCoMaxOwners) p.co_cap = kCoMaxOwners;
        const uint64_t bits = (uint64_t)p.co_wpg * p.L + p.co_margin;
        p.co_lds = ((bits / 32 + 8) + bits / 32 + ((bits >> p.co_shift) + 2) + 2 * ((size_t)p.co_wpg + 1)) * 4 + (size_t)p.co_cap * 8 + 64;
        if (bits + 0x1000 >= 0x40000 || p.co_lds + sizeof(CoParked) * kCoPark + 64 > 160 * 1024) p.co = 0;     // (k_hyp_walk_co packs distances inside a group into 19 bits)
    }
    {
        // (nodes of a chunk's first windows: room for one per 0.4 average coded data sets, as for the node records)
        uint64_t pn = cds * 2 / 5;
        if (pn < 8) pn = 8;
        if (pn > 64) pn = 64;
        const uint64_t wc = (uint64_t)p.wfirst * p.L / pn;
        p.wcap = (uint32_t)(wc < (1u << 20) ? wc : (1u << 20));
    }
    uint64_t look = 4 * (rsi_bits_hint ? rsi_bits_hint : (uint64_t)c.rsi * cds) + 8 * sync + 4 * p.L;
    if (look > (1ull << 27)) look = 1ull << 27;
    p.nlook = (uint32_t)((look + p.L - 1) / p.L);
    // nodes: one per 1.5 coded data sets on a trunk that is mostly off the true chain, one per coded data set on
    // it; room for 2.5 times the latter, never less than one per 64 bits
    uint64_t per_node = cds * 2 / 5;
    if (per_node < 8) per_node = 8;
    if (per_node > 64) per_node = 64;
    const uint64_t nwin_all = total_bits / p.L + 2;
    auto bytes_for = [&](uint64_t nwin) {
        p.ncap = (uint32_t)(nwin * p.L / per_node < 0xFFFFFF00ull ? nwin * p.L / per_node : 0xFFFFFF00ull);
        p.pcap = p.ncap / 2u + 65536u;               // (an RSI end per node that is no RSI end itself is plenty)
        return trunk_bytes(p, (uint32_t)nwin);
    };
    uint64_t nwin = nwin_all;
    size_t wanted = (size_t)(total_bits / 8u) * 6u;
    if (wanted < kTrWsWanted) wanted = kTrWsWanted;
    if (wanted > kTrWsMost) wanted = kTrWsMost;
    const size_t limit = ws_bytes ? ws_bytes : wanted;
    if (bytes_for(nwin) > limit) {
        // as many windows as fit (bytes are linear in nwin up to rounding)
        const size_t per = (bytes_for(4096) - bytes_for(2048)) / 2048 + 1;
        nwin = limit / per;
        while (nwin > p.nlook + 64 && bytes_for(nwin) > limit) nwin -= nwin / 64 + 1;
        if (nwin < (uint64_t)p.nlook * 2 + 64 || bytes_for(nwin) > limit) return p;      // not worth it: serial walk
    }
    if (nwin > 0x7FFFFFFFull / 2) return p;
    p.nwin_max = (uint32_t)nwin;
    p.bytes = bytes_for(nwin);
    p.ok = true;
    return p;
}

// Index pass over the trunk tables, span by span: trunk (count, repair, scan, fill), hypotheses (walk, jump,
// chain), wide walker, the walk itself (one wavefront: one lookup per chunk where the wide table resolves it,
// per window or per coded data set where not), then the RSI starts inside all hops.
void allow_big_lds_walk()
{
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [] {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_hyp_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_hyp_walk_co), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            (void)hipGetLastError();
    });
}

void launch_index_trunk(const Cfg &c, TrunkPlan p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                        uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                        uint8_t *base, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot, uint64_t *d_seg_bits,
                        const uint32_t *skip_if = nullptr)
{
    IdxCarry *carry = reinterpret_cast<IdxCarry *>(base);
    const TrStream s{words, nwords, end_bit};
    allow_big_lds_walk();
    (void)trunk_bytes(p, p.nwin_max);                     // offsets for the span size
    const uint64_t lo0 = start_bit / p.L * p.L;
    const uint64_t nwin_all = (end_bit - lo0) / p.L + 1;
    const bool one = nwin_all <= p.nwin_max;
    const uint32_t ncore_span = one ? (uint32_t)nwin_all : p.nwin_max - p.nlook;
    for (uint64_t lo = lo0; lo <= end_bit; lo += (uint64_t)ncore_span * p.L) {
        const uint64_t left = (end_bit - lo) / p.L + 1;
        const bool first = lo == lo0, last = left <= ncore_span || one;
        TrGeom g{};
        g.lo = lo;
        g.start_bit = start_bit;
        g.L = p.L;
        g.lead = p.lead;
        g.ncap = p.ncap;
        g.rw = p.rw;
        g.nwin = (uint32_t)(left < p.nwin_max ? left : p.nwin_max);
        g.ncore = (uint32_t)(left < ncore_span ? left : ncore_span);
        g.budget = p.budget;
        g.kmax = p.kmax;
        g.pcap = p.pcap;
        TrTables t{};
        t.bitmap = reinterpret_cast<uint32_t *>(base + p.o_bitmap);
        t.pre = reinterpret_cast<uint16_t *>(base + p.o_pre);
        t.nbase = reinterpret_cast<uint32_t *>(base + p.o_nbase);
        t.cpos = reinterpret_cast<uint16_t *>(base + p.o_cpos);
        t.bp = reinterpret_cast<uint32_t *>(base + p.o_bp);
        t.ccnt = reinterpret_cast<uint32_t *>(base + p.o_ccnt);
        t.nblk = reinterpret_cast<uint32_t *>(base + p.o_nblk);
        t.nros = reinterpret_cast<uint32_t *>(base + p.o_nros);
        t.entry = reinterpret_cast<uint64_t *>(base + p.o_entry);
        t.gbase = reinterpret_cast<uint64_t *>(base + p.o_gbase);
        t.seampre = reinterpret_cast<uint32_t *>(base + p.o_seam);
        t.rospre = reinterpret_cast<uint32_t *>(base + p.o_ros);
        t.rec = reinterpret_cast<TrRec *>(base + p.o_rec);
        t.park = reinterpret_cast<uint32_t *>(base + p.o_park);
        t.pool = reinterpret_cast<TrPoolEntry *>(base + p.o_pool);
        t.pool_cnt = reinterpret_cast<uint32_t *>(base + 48);       // (behind the carry record)
        t.skip_if = skip_if;
        uint64_t *ex[2] = {reinterpret_cast<uint64_t *>(base + p.o_exit0), reinterpret_cast<uint64_t *>(base + p.o_exit1)};
        const uint32_t nreg = (g.nwin + g.rw - 1) / g.rw;
        t.exit = ex[0];
        // (a wavefront per region where a coded data set fits the wavefront's register window, aec_coop.h; else a lane)
        // (... and where the regions are few: a wavefront's parse is mostly scalar work, and the one scalar unit of a CU
        // serves all its wavefronts -- measured on spans of 5167 regions: 4.5 ms against 6.1 for the lanes, whose time
        // is the latency of ONE region whatever their number)
        const bool coop = tune("AEC_TR_COOP", nreg <= 6144u ? 1u : 0u) && c.id_len + 1u + c.bps + c.bs * c.bps + 128u <= 2048u;
        const uint32_t cgrid = nreg < 256u * 20u ? nreg : 256u * 20u;
        if (coop)
            hipLaunchKernelGGL(k_trunk_coop, dim3(cgrid), dim3(64), 0, st, c, s, g, t, (const uint64_t *)nullptr, ex[0], 0u);
        else
            hipLaunchKernelGGL(k_trunk, dim3((nreg + 63) / 64), dim3(64), 0, st, c, s, g, t, (const uint64_t *)nullptr, ex[0], 0u);
        uint32_t cur = 0;
        for (uint32_t k = 0; k < p.passes; k++) {
            if (coop)
                hipLaunchKernelGGL(k_trunk_coop, dim3(cgrid), dim3(64), 0, st, c, s, g, t, (const uint64_t *)ex[cur],
                                   ex[cur ^ 1u], 1u);
            else
                hipLaunchKernelGGL(k_trunk, dim3((nreg + 63) / 64), dim3(64), 0, st, c, s, g, t, (const uint64_t *)ex[cur],
                                   ex[cur ^ 1u], 1u);
            cur ^= 1u;
        }
        t.exit = ex[cur];
        hipLaunchKernelGGL(k_trunk_scan, dim3(1), dim3(1024), 0, st, g, t);
        hipLaunchKernelGGL(k_trunk, dim3((g.nwin + 63) / 64), dim3(64), 0, st, c, s, g, t, (const uint64_t *)nullptr,
                           (uint64_t *)nullptr, 2u);
        (void)hipMemsetAsync(t.pool_cnt, 0, 4, st);
        const uint32_t ngroups = (g.ncore + p.wpw - 1) / p.wpw;
        CoLists ls{};
        if (p.co) {
            ls.queue = reinterpret_cast<uint2 *>(base + p.o_coq);
            ls.plain = reinterpret_cast<uint2 *>(base + p.o_cop);
            ls.counts = reinterpret_cast<uint32_t *>(base + 56);    // (behind the carry record and the pool counter)
            ls.qcap = p.co_qcap;
            ls.pcap = p.co_pcap;
            ls.over_plain = p.co_over;
            (void)hipMemsetAsync(ls.counts, 0, 8, st);
            const uint32_t ng = (g.ncore + p.co_wpg - 1) / p.co_wpg;
            const size_t lds_wg = p.co_lds + sizeof(CoParked) * kCoPark + 64;          // (dynamic + the kernel's own)
            const uint32_t per_cu = (uint32_t)(160 * 1024 / lds_wg) ? (uint32_t)(160 * 1024 / lds_wg) : 1u;
            const uint32_t grid = ng < 256u * per_cu ? ng : 256u * per_cu;
            hipLaunchKernelGGL(k_hyp_walk_co, dim3(grid), dim3(256), p.co_lds, st, c, s, g, t, p.co_wpg, p.co_margin, ng,
                               p.co_shift, p.co_tmax, p.co_cap, ls, p.co_park);
            hipLaunchKernelGGL(k_hyp_walk_rest, dim3((p.co_qcap + 63) / 64 < 16384u ? (p.co_qcap + 63) / 64 : 16384u), dim3(64), 0,
                               st, c, s, g, t, ls);
            hipLaunchKernelGGL(k_hyp_defer, dim3(g.ncore), dim3(256), 0, st, c, g, t, ls);
        } else if (p.staged) {
            const uint32_t grid = ngroups < 256u ? ngroups : 256u;                           // (one workgroup per CU)
            hipLaunchKernelGGL((k_hyp_walk<true>), dim3(grid), dim3(1024), p.lds, st, c, s, g, t, p.wpw, p.margin, ngroups);
        } else {
            hipLaunchKernelGGL(k_hyp_walk_mem, dim3(ngroups), dim3(64), 0, st, c, s, g, t, p.wpw);
        }
        hipLaunchKernelGGL(k_hyp_land, dim3(g.ncore), dim3(256), 0, st, c, g, t);
        if (p.co)
            hipLaunchKernelGGL(k_hyp_walk_list, dim3(p.co_pcap < 2048u ? p.co_pcap : 2048u), dim3(64), 0, st, c, s, g, t, ls);

        const uint32_t nwin = g.ncore, nchunks = (nwin + p.wpc - 1) / p.wpc;
        TwTables sp;
        sp.bitmap = t.bitmap;
        sp.pre = t.pre;
        sp.nbase = t.nbase;
        sp.ccnt = t.ccnt;
        sp.rec = t.rec;
        sp.cpos = t.cpos;
        sp.lo = lo;
        sp.hi = lo + (uint64_t)nwin * p.L;
        sp.core = p.L;
        sp.wide = reinterpret_cast<const uint4 *>(base + p.o_wide);
        sp.wpc = p.wpc;
        sp.wcap = p.wcap;
        sp.wfirst = p.wfirst;
        sp.nrec = p.ncap;
        sp.skip_if = skip_if;
        ChunkEntry *centry = reinterpret_cast<ChunkEntry *>(base + p.o_centry);
        IdxHop *hops = reinterpret_cast<IdxHop *>(base + p.o_hops);
        const uint32_t hop_cap = 2 * nwin + 8;
        (void)hipMemsetAsync(centry, 0, (size_t)nchunks * sizeof(ChunkEntry), st);
        hipLaunchKernelGGL(k_twide, dim3((p.wcap + 255) / 256, nchunks), dim3(256), 0, st, sp, nwin, end_bit,
                           const_cast<uint4 *>(sp.wide));
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi,
                           d_res, (const uint64_t *)nullptr, hops, hop_cap, carry, first ? 1u : 0u, last ? 1u : 0u,
                           start_block, rsi_start, tail_slot, sp, centry, SparseTables{}, (uint32_t *)nullptr, (uint64_t)0,
                           skip_if);
        hipLaunchKernelGGL(k_trewalk, dim3((nchunks + 63) / 64), dim3(64), 0, st, c, sp, t, nwin, nchunks, end_bit, centry,
                           d_rsi_off);
        hipLaunchKernelGGL(k_texpand, dim3((hop_cap + 255) / 256), dim3(256), 0, st, c, sp, t, carry, hops, d_rsi_off);
        if (d_seg_bits) {
            // (at most one RSI per minimal coded RSI of the span; a wavefront each, the rest in turn)
            const uint64_t most = (uint64_t)g.ncore * g.L / ((uint64_t)c.segs_per_rsi * (c.id_len + 2u)) + 2u;
            const uint32_t gmax = tune("AEC_TR_SEG_GRID", 4096u);
            const uint32_t grid = (uint32_t)(most < gmax ? most : gmax);
            hipLaunchKernelGGL(k_seg_starts, dim3(grid), dim3(64), 0, st, c, s, g, t, d_rsi_off, carry, d_res, d_seg_bits,
                               max_rsi + 1u);
        }
        if (last) break;
    }
#ifdef AEC_TUNING
    if (tune_set("AEC_IDX_STATS")) {                       // (diagnostics: synchronises)
        IdxCarry h{};
        (void)hipStreamSynchronize(st);
        const int dbg = tune("AEC_IDX_STATS", 0) >= 2 ? 1 : 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_serial), &dbg, sizeof(dbg));
        (void)hipMemcpy(&h, carry, sizeof(h), hipMemcpyDeviceToHost);
        uint32_t co_counts[2] = {0, 0};
        if (p.co) (void)hipMemcpy(co_counts, base + 56, 8, hipMemcpyDeviceToHost);
        fprintf(stderr, "trunk index: L %u lead %u rw %u kmax %u wpw %u nwin/span %u ncap %u | RSIs %llu, walked serially %u, "
                "table lookups %u | coalescing %u (groups of %u windows + %u bits, %u-bit cells, %u parses): last span handed on "
                "%u walks, left %u nodes to the plain walk\n", p.L, p.lead, p.rw, p.kmax, p.wpw, p.nwin_max, p.ncap,
                (unsigned long long)h.r, h.n_serial, h.n_lookups, p.co, p.co_wpg, p.co_margin, 1u << p.co_shift, p.co_tmax,
                co_counts[0], co_counts[1]);
    }
#endif
}

}  // namespace

// does the index pass of such a stream run over the window tables (cheap per call, also for small pieces)?
bool index_is_windowed(const Cfg &c, size_t in_bytes, uint64_t rsi_bits_hint)
{
    return in_bytes && sparse2_plan(c, (uint64_t)in_bytes * 8, rsi_bits_hint).ok;
}

namespace {

// ======== short RSIs: phase-locked chains =================================================================
// Both table schemes follow chains that parse WITHOUT reference samples and try RSI starts as hypotheses on them.
// That needs RSIs much longer than the distance such a chain takes to find the true one again behind an RSI start
// (a hundred coded data sets): with RSIs of 1 .. 32 blocks -- narrow SZIP scan lines -- no chain is ever on the true
// one, every RSI fell to the serial walker, and 16 MiB took 0.7 s (one host core with the reference: 0.03 s).
// Short RSIs allow something simpler: a chain that keeps the RSI's bookkeeping itself (a reference sample every `rsi`
// blocks, zero runs by the block count) is the TRUE chain for good once it stands on an RSI start with its count at
// zero -- it locks, after about (bits per coded data set) x rsi steps from anywhere.  So: the stream in regions;
// per region a wavefront of 64 such chains from 64 places in front of it, the state most of them agree on where the
// region begins is the guess for its entry; one lane walks the region from there (count); regions whose entry is
// not where the region in front of them ended are walked again from there (repair passes, as for the trunk; the
// first region starts from the caller's exact state); a scan numbers the RSIs; a last walk writes their starts.
// If the regions do not all agree in the end, or a coded data set of the agreed chain does not parse, nothing is
// delivered and the serial walker behind takes the stream as before: results never rest on the guesses.
struct LkState {
    uint64_t pos;
    uint32_t b, st;        // blocks of the current RSI done; st: 0 walking, 1 no coded data set ends inside the input, 2 refused
};
struct LockTables {
    LkState *entry, *exit0, *exit1;
    uint32_t *cnt;         // RSI starts met in the region
    uint64_t *base;        // ... in front of the region
    uint32_t *flags;       // [0] done (k_index behind returns at once), [1] inconsistent, [2] region that ended the walk,
                           // [3] abandoned (plausibility guesses: too many did not hold; the trunk takes the stream)
    uint32_t nreg, region_bits, lead;
    uint32_t gap;          // bits between the starts of the 64 chains of a guess (k_lock_guess_w)
    uint32_t coop;         // long coded data sets: the walks parse one at a time (lk_walk_coop) instead of 64 bits at a time
    uint32_t mode;         // 1: the entries are guesses by plausibility (walks from wrong ones do not lock by themselves)
    const uint32_t *skip_if;   // != 0 there: a scheme in front has delivered the stream; every kernel returns at once
    uint64_t lo;           // bit position where region 0 begins (the caller's start)
};

__device__ __forceinline__ void lk_step(const TrStream &s, const Cfg &c, LkState &x)
{
    const uint32_t ref = (x.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u;
    uint32_t nz;
    const uint32_t len = tr_cds(s, c, x.pos, ref, nz);
    if (!len) {
        x.st = 1u;
        return;
    }
    const uint32_t nb = tr_blocks(c, nz, x.b);
    if (!nb) {
        x.st = 2u;
        return;
    }
    x.pos += len;
    x.b += nb;
    if (x.b >= c.rsi) x.b = 0u;
}

// the guess for the entry of every region but the first: 64 chains from 64 places in front of it
__global__ void __launch_bounds__(64)
k_lock_guess(const Cfg c, const TrStream s, const LockTables t)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t r = 1u + blockIdx.x; r < t.nreg; r += gridDim.x) {
        const uint64_t rstart = t.lo + (uint64_t)r * t.region_bits;
        uint64_t from = rstart > t.lo + t.lead ? rstart - t.lead : t.lo;
        LkState x{from + (uint64_t)lane * 37u, 0u, 0u};
        // (a chain that cannot go on starts again one bit further: it is a guess either way)
        uint32_t steps = 0;
        const uint32_t most = 4u * (t.lead / (c.id_len + 1u) + 64u);
        while (x.pos < rstart && steps++ < most) {
            const uint32_t b_was = x.b;
            lk_step(s, c, x);
            if (x.st) {
                if (x.st == 1u) break;
                // (a zero run that does not fit the RSI by this chain's count: the count is wrong -- an RSI may start here;
                // refused with the count at zero as well: one bit on)
                if (b_was == 0u) x.pos++;
                x.b = 0u;
                x.st = 0u;
            }
        }
        const bool have = x.st == 0u && x.pos >= rstart;
        // the state most lanes stand on
        uint64_t left = __ballot(have);
        LkState best{rstart, 0u, 0u};
        uint32_t best_n = 0;
        for (uint32_t round = 0; round < 8u && left; round++) {
            const uint32_t l0 = (uint32_t)__builtin_ctzll(left);
            const uint64_t p0 = __shfl(x.pos, (int)l0);
            const uint32_t b0 = (uint32_t)__shfl((int)x.b, (int)l0);
            const uint64_t same = __ballot(have && x.pos == p0 && x.b == b0);
            const uint32_t n = (uint32_t)__popcll(same);
            if (n > best_n) {
                best_n = n;
                best = LkState{p0, b0, 0u};
            }
            left &= ~same;
        }
        if (lane == 0) t.entry[r] = best;
    }
}

// mode 0: count from the guessed entries; 1: repair (regions whose entry is not the exit of the region in front)
__global__ void __launch_bounds__(64)
k_lock_walk(const Cfg c, const TrStream s, const LockTables t, const LkState *exit_prev, LkState *exit_out, uint32_t mode,
            uint64_t start_bit, uint32_t start_block)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nreg) return;
    LkState x;
    if (r == 0u) {
        if (mode) {
            exit_out[0] = exit_prev[0];
            return;
        }
        x = LkState{start_bit, start_block, 0u};
        t.entry[0] = x;
    } else if (!mode) {
        x = t.entry[r];
    } else {
        const LkState prev = exit_prev[r - 1u], mine = t.entry[r];
        if (prev.st || (prev.pos == mine.pos && prev.b == mine.b)) {
            exit_out[r] = exit_prev[r];
            return;
        }
        x = LkState{prev.pos, prev.b, 0u};
        t.entry[r] = x;
    }
    // (the last region has no end: its walk stops where the input does)
    const uint64_t rend = r + 1u == t.nreg ? ~0ull : t.lo + (uint64_t)(r + 1u) * t.region_bits;
    uint32_t n = 0;
    while (x.pos < rend) {
        n += x.b == 0u ? 1u : 0u;
        lk_step(s, c, x);
        if (x.st) break;
    }
    t.cnt[r] = n;
    exit_out[r] = x;
}

// one workgroup: the first region in which the walk ended (those behind it are dead: nobody confirmed their
// guesses), do the live regions agree, RSI starts in front of every region
__global__ void __launch_bounds__(1024)
k_lock_scan(const LockTables t, const LkState *exit_last)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ uint64_t sh[1024];
    __shared__ uint32_t bad, first_end;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) {
        bad = 0u;
        first_end = t.nreg - 1u;
    }
    __syncthreads();
    const uint32_t per = (t.nreg + nt - 1u) / nt, lo = tid * per, hi = lo + per < t.nreg ? lo + per : t.nreg;
    for (uint32_t r = lo; r < hi; r++)
        if (exit_last[r].st) {
            atomicMin(&first_end, r);
            break;
        }
    __syncthreads();
    const uint32_t fe = first_end;
    uint64_t sum = 0;
    uint32_t wrong = 0;
    for (uint32_t r = lo; r < hi; r++) {
        const bool live = r <= fe;
        if (live && r) {
            const LkState prev = exit_last[r - 1u], mine = t.entry[r];
            if (prev.pos != mine.pos || prev.b != mine.b) wrong = 1u;
        }
        if (!live) t.cnt[r] = 0u;
        sum += live ? t.cnt[r] : 0u;
    }
    sh[tid] = sum;
    if (wrong) atomicOr(&bad, 1u);
    __syncthreads();
    if (tid == 0) {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < nt; i++) {
            const uint64_t v = sh[i];
            sh[i] = acc;
            acc += v;
        }
    }
    __syncthreads();
    uint64_t acc = sh[tid];
    for (uint32_t r = lo; r < hi; r++) {
        t.base[r] = acc;
        acc += t.cnt[r];
    }
    if (tid == 0) {
        t.flags[1] = bad;
        t.flags[2] = fe;
    }
}

// the RSI starts, and the result record; the first region in which the walk ended writes the record
__global__ void __launch_bounds__(64)
k_lock_fill(const Cfg c, const TrStream s, const LockTables t, const LkState *exit_last, const uint32_t *__restrict__ words,
            uint64_t nwords, uint64_t *__restrict__ rsi_off, uint64_t max_rsi, DecResult *res, uint32_t tail_slot,
            uint64_t rsi_start_in, uint32_t start_block)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nreg || t.flags[1] || r > t.flags[2]) return;      // (not agreed, or behind the region that ended the walk)
    const uint64_t rend = r + 1u == t.nreg ? ~0ull : t.lo + (uint64_t)(r + 1u) * t.region_bits;
    LkState x = t.entry[r];
    // A walk that resumes INSIDE an RSI (streaming callers: start_block blocks of it lie in front of the input): that
    // RSI is number 0 and began at rsi_start_in, the first RSI start the chains meet is number 1 (as k_index counts).
    const uint64_t off = start_block ? 1u : 0u;
    uint64_t idx = t.base[r] + off;
    if (off && r == 0u && max_rsi) rsi_off[0] = rsi_start_in;
    // (regions behind the caller's bound have nothing to deliver: the ONE region in which RSI number max_rsi starts
    // writes the record -- every region behind it would meet "idx >= max_rsi" at its first RSI start as well)
    if (idx > max_rsi) return;
    // the last RSI start in front of this region (the RSI the walk is in when it enters): found by walking the nearest
    // region in front that met one -- only the lane that ends the walk asks
    auto start_in_front = [&]() -> uint64_t {
        for (uint32_t q = r; q-- > 0u;) {
            if (!t.cnt[q]) continue;
            LkState y = t.entry[q];
            const uint64_t qend = t.lo + (uint64_t)(q + 1u) * t.region_bits;
            uint64_t last = rsi_start_in;
            while (y.pos < qend) {
                if (y.b == 0u) last = y.pos;
                lk_step(s, c, y);
                if (y.st) break;
            }
            return last;
        }
        return rsi_start_in;
    };
    uint64_t cur = 0;                                    // start of the RSI the walk is in, if it began in this region
    bool met = false, clipped = false;
    while (x.pos < rend) {
        if (x.b == 0u) {
            if (idx == max_rsi) {                        // the caller's bound: ends on this RSI start
                clipped = true;
                break;
            }
            rsi_off[idx] = x.pos;
            cur = x.pos;
            met = true;
            idx++;
        }
        lk_step(s, c, x);
        if (x.st) break;
    }
    if (!clipped && !x.st) return;                       // the walk goes on in the next region
    if (clipped) {
        // (only the region that holds RSI number max_rsi gets here with idx == max_rsi at an RSI start)
        res->n_rsi = max_rsi;
        res->tail_blocks = 0;
        res->end_bit = x.pos;
        res->status = DEC_OK;                            // (the whole record: the walker behind does not start it then)
        res->pad = 0u;
        res->bad_rsi = ~0ull;
        if (tail_slot) rsi_off[max_rsi] = met ? cur : start_in_front();
        __threadfence();
        t.flags[0] = 1u;
        return;
    }
    // the walk ended here: only "the input ends inside this coded data set", confirmed by the sequential reader, is
    // delivered; anything else is the serial walker's to report
    if (x.st != 1u) return;
    // (st 1 is also what a unary part beyond the parser's reach reports -- a foreign encoder's coded data set of 12 800
    // bits: only within that reach of the end of the input does it mean that the input ended)
    if (s.end_bit - x.pos > kTrMaxScan) return;
    {
        BitReaderT<QuadFetch> br;
        br.init(QuadFetch{words, nwords}, s.end_bit, x.pos);
        uint32_t nblk = 1;
        if (skip_cds(br, c, (x.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u, x.b, nblk) != DEC_NEED_INPUT) return;
    }
    // RSI starts met = idx; the last of them began an RSI that is not complete (or nothing at all behind it)
    if (idx == 0) return;                                // (cannot happen: the first state is an RSI start)
    res->n_rsi = idx - 1u;
    res->tail_blocks = x.b;
    res->end_bit = x.pos;
    res->status = DEC_OK;
    res->pad = 1u;
    res->bad_rsi = ~0ull;
    if (tail_slot) rsi_off[max_rsi] = met ? cur : start_in_front();
    __threadfence();
    t.flags[0] = 1u;
}

// ---- the same three walks, a WAVEFRONT per region (round 5; aec_stretch.h) ---------------------------------------
// A lane that follows a chain through device memory takes 2.5 us per coded data set, and the pass was three such walks
// of a region one behind the other (guess, count, fill) plus a repair pass per wrong guess: 2.1 ms for a 64 KiB chunk
// with scan lines of 32 pixels, 29 ms for 16 MiB of 16-bit data with rsi 32 -- as long as the reference takes on one
// core.  Here a wavefront owns the region: the 64 lanes parse the coded data sets that would begin at the next 64 bits,
// with and without a reference sample, and the chain with the RSI's bookkeeping is a lane read per coded data set.  The
// guess keeps its 64 chains per region, one per lane, but they parse out of the wavefront's LDS tables (they stand
// within a few kbit of each other) instead of device memory.  Same states, same tables, same results as the kernels
// above (tune AEC_IDX_LOCK_WAVE=0 runs those).

// one step of the bookkeeping from an entry; false = the entry does not resolve it (lk_step from memory then)
__device__ __forceinline__ bool lk_apply(const Cfg &c, LkState &x, uint32_t e, uint32_t rf)
{
    if (!(e >> 12)) return false;
    const uint32_t len = e & 0xFFFu;
    uint32_t nb = 1;
    if (e & kNxtZero) {
        nb = spec_run_blocks(c, len - c.id_len - 1u - rf * c.bps, x.b);
        if (!nb) {
            x.st = 2u;
            return true;
        }
    }
    x.pos += len;
    x.b += nb;
    if (x.b >= c.rsi) x.b = 0u;
    return true;
}

// The chain from x (wave-uniform) until x.pos >= rend or x.st != 0; at_start(x) is called once at every step that
// begins an RSI (x.b == 0), before the step; it returns false to end the walk there.
template <class WS, class F>
__device__ __forceinline__ void lk_walk_wave(WS &ws, const TrStream &s, const Cfg &c, LkState &x, uint64_t rend,
                                             F at_start)
{
    const bool pp = c.flags & F_PREPROCESS;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t handled = ~0ull;
    while (x.pos < rend && !x.st) {
        const uint32_t rel = ws.ensure(x.pos, 64u);
        // (the entries with a reference sample only where an RSI starts inside these 64 bits)
        uint32_t E0 = 0, E1 = 0;
        bool have0 = false, have1 = false;
        const uint64_t p0 = x.pos;
        bool moved = false;
        while (x.pos < rend && x.pos - p0 < 64u) {
            if (x.b == 0u && handled != x.pos) {
                handled = x.pos;
                if (!at_start(x)) return;
            }
            const uint32_t rf = (pp && x.b == 0u) ? 1u : 0u;
            if (rf && !have1) {
                E1 = ws.entry(c, rel + lane, 1u);
                have1 = true;
            } else if (!rf && !have0) {
                E0 = ws.entry(c, rel + lane, 0u);
                have0 = true;
            }
            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x.pos - p0));
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)(rf ? E1 : E0), (int)l);
            if (!lk_apply(c, x, e, rf)) break;
            moved = true;
            if (x.st) return;
        }
        if (!moved && x.pos < rend) {                   // (not resolved by the tables: from memory, as before)
            if (x.b == 0u && handled != x.pos) {
                handled = x.pos;
                if (!at_start(x)) return;
            }
            lk_step(s, c, x);
        }
    }
}

// The same walk where coded data sets are long (hundreds of bits: one boundary in ten such steps): one coded data set at
// a time, parsed by the wavefront out of the window alone (aec_stretch.h: coop_half) -- 0.25 us each, where the piece
// tables cost 7 us per 2048 bits to look up the three boundaries in them.
template <class WS, class F>
__device__ __forceinline__ void lk_walk_coop(WS &ws, const TrStream &s, const Cfg &c, LkState &x, uint64_t rend,
                                             F at_start)
{
    const bool pp = c.flags & F_PREPROCESS;
    uint64_t handled = ~0ull;
    while (x.pos < rend && !x.st) {
        if (x.b == 0u && handled != x.pos) {
            handled = x.pos;
            if (!at_start(x)) return;
        }
        const uint32_t rf = (pp && x.b == 0u) ? 1u : 0u;
        const uint32_t rel = ws.ensure_win(x.pos, 64u);
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)ws.coop_half(c, rel, rf));
        if (!lk_apply(c, x, e, rf)) lk_step(s, c, x);
    }
}
template <class WS, class F>
__device__ __forceinline__ void lk_walk_any(bool coop, WS &ws, const TrStream &s, const Cfg &c, LkState &x, uint64_t rend,
                                            F at_start)
{
    if (coop)
        lk_walk_coop(ws, s, c, x, rend, at_start);
    else
        lk_walk_wave(ws, s, c, x, rend, at_start);
}

__global__ void __launch_bounds__(256)
k_lock_guess_w(const Cfg c, const TrStream s, const LockTables t)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t lk_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveStream ws;
    ws.init(lk_lds + (size_t)wave * kSwWaveWords, s, c);
    const bool pp = c.flags & F_PREPROCESS;
    const uint32_t r = 1u + blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= t.nreg) return;
    const uint64_t rstart = t.lo + (uint64_t)r * t.region_bits;
    const uint64_t from = rstart > t.lo + t.lead ? rstart - t.lead : t.lo;
    LkState x{from + (uint64_t)lane * t.gap, 0u, 0u};
    uint32_t steps = 0;
    const uint32_t most = 4u * (t.lead / (c.id_len + 1u) + 64u);
    bool ended = false;                                 // (no coded data set ends inside the input from here: out)
    for (;;) {
        const bool run = !ended && x.pos < rstart && steps < most;
        if (!__any(run)) break;
        // the chains that lag furthest: window and piece go where they stand, the others wait
        uint64_t lo = run ? x.pos : ~0ull;
        for (int off = 32; off; off >>= 1) {
            const uint64_t o = __shfl_xor(lo, off);
            lo = o < lo ? o : lo;
        }
        const uint32_t rel0 = ws.ensure(lo, kSwPiece / 2u);
        const uint64_t pend = ws.base * 32u + ws.pc0 + kSwPiece, pbeg = ws.base * 32u + ws.pc0;
        (void)rel0;
        // (a lane outside the piece waits; every round moves the laggards at least half a piece on)
        for (uint32_t it = 0; it < 4096u; it++) {
            const bool go = !ended && x.pos < rstart && steps < most && x.pos >= pbeg && x.pos < pend &&
                            x.pos < lo + kSwPiece / 2u;
            if (!__any(go)) break;
            if (go) {
                steps++;
                const uint32_t b_was = x.b;
                const uint32_t rf = (pp && x.b == 0u) ? 1u : 0u;
                const uint32_t e = ws.entry(c, (uint32_t)(x.pos - ws.base * 32u), rf);
                if (!lk_apply(c, x, e, rf)) lk_step(s, c, x);
                if (x.st) {
                    if (x.st == 1u) {
                        ended = true;
                    } else {
                        if (b_was == 0u) x.pos++;
                        x.b = 0u;
                        x.st = 0u;
                    }
                }
            }
        }
    }
    const bool have = !ended && x.st == 0u && x.pos >= rstart;
    uint64_t left = __ballot(have);
    LkState best{rstart, 0u, 0u};
    uint32_t best_n = 0;
    for (uint32_t round = 0; round < 8u && left; round++) {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(left);
        const uint64_t p0 = __shfl(x.pos, (int)l0);
        const uint32_t b0 = (uint32_t)__shfl((int)x.b, (int)l0);
        const uint64_t same = __ballot(have && x.pos == p0 && x.b == b0);
        const uint32_t n = (uint32_t)__popcll(same);
        if (n > best_n) {
            best_n = n;
            best = LkState{p0, b0, 0u};
        }
        left &= ~same;
    }
    if (lane == 0) t.entry[r] = best;
}

__global__ void __launch_bounds__(256)
k_lock_walk_w(const Cfg c, const TrStream s, const LockTables t, const LkState *exit_prev, LkState *exit_out, LkState *entry_out,
              uint32_t mode, uint64_t start_bit, uint32_t start_block)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t lk_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= t.nreg || t.flags[3]) return;
    LkState x;
    if (r == 0u) {
        if (mode) {
            if (lane == 0) {
                exit_out[0] = exit_prev[0];
                entry_out[0] = t.entry[0];
            }
            return;
        }
        x = LkState{start_bit, start_block, 0u};
        if (lane == 0) entry_out[0] = x;
    } else if (!mode) {
        x = t.entry[r];
    } else {
        // A region whose entry is not the exit in front is walked again from that exit -- if the region in front is not
        // in doubt itself: the exit of a walk from a wrong entry is wrong too (the position finds the true chain again,
        // the count of blocks does not), and walking on from it would throw a right guess after a wrong one, region by
        // region down the stream.  The first region of a run in doubt always has an undoubted one in front.
        const LkState prev = exit_prev[r - 1u], mine = t.entry[r];
        bool idle = prev.st || (prev.pos == mine.pos && prev.b == mine.b);
        // (Where 64 chains lock onto the phase -- short RSIs -- a walk from a wrong entry locks as well and its exit is
        // mostly right: there every region in doubt is walked again at once.)
        if (!idle && r >= 2u && t.mode == 1u) {
            const LkState pp = exit_prev[r - 2u], pe = t.entry[r - 1u];
            idle = !pp.st && (pp.pos != pe.pos || pp.b != pe.b);
        }
        // (the wavefront of region r + 1 reads this region's entry in the same pass: entries have a twin like the exits)
        if (lane == 0) entry_out[r] = idle ? mine : LkState{prev.pos, prev.b, 0u};
        if (idle) {
            if (lane == 0) exit_out[r] = exit_prev[r];
            return;
        }
        x = LkState{prev.pos, prev.b, 0u};
    }
    const uint64_t rend = r + 1u == t.nreg ? ~0ull : t.lo + (uint64_t)(r + 1u) * t.region_bits;
    WaveStream ws;
    ws.init(lk_lds + (size_t)wave * kSwWaveWords, s, c);
    uint32_t n = 0;
    if (ws.usable()) {
        lk_walk_any(t.coop != 0u, ws, s, c, x, rend, [&](const LkState &) {
            n++;
            return true;
        });
    } else {
        while (x.pos < rend) {
            n += x.b == 0u ? 1u : 0u;
            lk_step(s, c, x);
            if (x.st) break;
        }
    }
    if (lane == 0) {
        t.cnt[r] = n;
        exit_out[r] = x;
    }
}

// ---- entries by PLAUSIBILITY (round 5): streams of long coded data sets in RSIs too short for any chain ---------------
// The reference's own sample shape (16-bit, blocks of 64, rsi 256): coded data sets of 720 bits, so a chain needs ~2 cds^2 =
// 1 Mbit to fall onto the true one -- and an RSI is 184 kbit: between two reference samples no chain that parses without
// them ever gets there.  The trunk is mostly off the true chain (17 % of the RSI starts are nodes), its hypotheses walk
// 1500 coded data sets each, and the index ran at 9.7 GB/s; the phase-locked chains would need lead-ins of 265 Mbit.
// What such a stream does have: consecutive blocks of real data take NEIGHBOURING code options, while a parse from a wrong
// bit reads its option out of payload bits -- random.  So the entry of a region is GUESSED from that:
//   1. every bit of a stretch as long as the longest coded data set (one of them is a boundary) starts a chain of 12
//      coded data sets without reference samples, a lane each; the chain most of whose options lie within one of their
//      predecessor's stands on true boundaries (a wrong one that scores as well has met the true chain on its way);
//   2. from there a wavefront walks on, 64 bits per step as everywhere in this scheme, until an option does NOT fit its
//      predecessor: the coded data set in front of it was parsed without the reference sample it has -- an RSI starts
//      there -- if the parse WITH one is followed by plausible coded data sets again (else the data jumped: on it goes);
//   3. from that RSI start the walk with the RSI's bookkeeping is exact up to the region's start: the entry.
// Nothing rests on the guess: entries are checked against the exits of the regions in front, repaired and mended by the
// kernels of the phase-locked scheme; k_lock_judge counts the guesses that did not hold after the first walk and, if they
// are many (data whose options say nothing: noise in every block, periodic streams), hands the stream to the trunk.
// RSIs of up to this many blocks take the phase-locked scheme.  32 until round 5: the window tables resolve RSIs that
// short badly -- 16 MiB of 8-bit data with rsi 33: 72 ms, more than the reference on one core, rsi 40: 17 ms (16-bit: 18
// and 6.7) -- and win from ~48 blocks on (5.5 / 3.1 ms; the chains' lead-in grows with rsi): tests/bench_short_rsi.py --edges
constexpr uint32_t kLockMaxRsi = 44;
constexpr uint32_t kLpSteps = 16;          // coded data sets per scoring chain
constexpr uint32_t kLpAccept = 8;          // options within 1 of their predecessor's (of kLpSteps - 1) that make a chain the true one
constexpr uint32_t kLpConfirm = 8;         // coded data sets looked at where the walk doubts ...
constexpr uint32_t kLpConfirmOk = 6;       // ... and how many of their options must lie within 2 of their predecessor's
constexpr uint32_t kLpPhased = 16;         // RSIs of fewer blocks: the scoring chains carry the count of blocks (find_anchor_phased)
constexpr uint32_t kLpLost = 3;            // fewer than this along the plain chain: the walk has lost the true one
constexpr uint32_t kLpSpan = 10240;        // bits of the window in front of the walk (a confirmation's coded data sets)
// (tests/emul-style check on the reference's sample file, 120 region starts: 116 entries right, 2 without an anchor in a
// noisy stretch, 2 lost next to the start of the stream; with 12 steps, options within 1 and all of 5 confirmations: 22 of 31)

__device__ __forceinline__ uint32_t lp_id(const TrStream &s, const Cfg &c, uint64_t q)
{
    return (uint32_t)(tr_peek64(s, q) >> (64u - c.id_len));
}

#ifdef AEC_TUNING
__device__ unsigned long long g_lp_prof[16];       // (diagnostics, AEC_IDX_STATS=1: shader-clock ticks and counts of the guesses, summed)
#define LP_ADD(k, v) do { if (lane == 0) atomicAdd(&g_lp_prof[k], (unsigned long long)(v)); } while (0)
#define LP_MAX(k, v) do { if (lane == 0) atomicMax(&g_lp_prof[k], (unsigned long long)(v)); } while (0)
#define LP_NOW() __builtin_amdgcn_s_memtime()
#else
#define LP_MAX(k, v) do { } while (0)
#define LP_ADD(k, v) do { } while (0)
#define LP_NOW() 0ull
#endif
constexpr uint32_t kLpPiece = 16384;       // the guess's piece of the stream: 16 coded data sets of a scoring chain fit it

__global__ void __launch_bounds__(64)
k_lock_guess_p(const Cfg c, const TrStream s, const LockTables t, uint32_t back_bits)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t lk_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = 1u + blockIdx.x;
    if (r >= t.nreg) return;
    typedef WaveStreamT<kLpPiece> WS;
    WS ws;
    ws.init(lk_lds, s, c);
    const bool fast = ws.usable();
    const bool pp = c.flags & F_PREPROCESS;
    const uint64_t rstart = t.lo + (uint64_t)r * t.region_bits;
    const uint64_t A = rstart > t.lo + back_bits ? rstart - back_bits : t.lo;
    const uint32_t maxbits = c.id_len + 1u + c.bps + c.bs * c.bps;
    auto near1 = [](uint32_t a, uint32_t b) { return (a > b ? a - b : b - a) <= 1u; };
    auto near2 = [](uint32_t a, uint32_t b) { return (a > b ? a - b : b - a) <= 2u; };
    // ---- 1. a position on the true chain: every bit of a stretch as long as the longest coded data set starts a chain,
    // a lane each; returns where the best one stands after three coded data sets (0: none is convincing)
    // (the lanes parse at their own positions out of the wavefront's LDS tables -- the piece holds the 16 coded data sets
    // of a chain; from device memory where a chain leaves it: 2.5 us instead of 0.3 per coded data set)
    auto find_anchor = [&](uint64_t from) -> uint64_t {
        uint32_t best_sc = 0;
        uint64_t best_at = 0;
        uint64_t pbeg = 0, pend = 0;
        if (fast) {
            (void)ws.ensure(from, WS::kPiece / 2u);
            pbeg = ws.base * 32u + ws.pc0;
            pend = pbeg + WS::kPiece;
        }
        for (uint64_t q0 = from; q0 < from + maxbits + 64u; q0 += 64u) {
            uint64_t q = q0 + lane, at3 = 0;
            uint32_t sc = 0, prev = 0;
            bool ok = true;
            for (uint32_t i = 0; i < kLpSteps && ok; i++) {
                uint32_t nz, len = 0, id;
                if (q >= pbeg && q < pend) {
                    const uint32_t rel = (uint32_t)(q - ws.base * 32u);
                    len = ws.entry(c, rel, 0u) & 0xFFFu;
                    id = ws.option(c, rel);
                } else {
                    id = lp_id(s, c, q);
                }
                if (!len) len = tr_cds(s, c, q, 0u, nz);
                ok = len != 0u;
                if (i && near1(id, prev)) sc++;
                prev = id;
                if (i == 3u) at3 = q;
                q += len;
            }
            if (!ok) sc = 0;
            if (sc > best_sc) {
                best_sc = sc;
                best_at = at3;
            }
        }
        uint32_t top = best_sc;
        for (int off = 32; off; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)top, off);
            top = o > top ? o : top;
        }
        const uint64_t who = __ballot(best_sc == top);
        const uint64_t at = __shfl(best_at, (int)__builtin_ctzll(who));
        return top >= kLpAccept ? at : 0ull;
    };
    // The same for RSIs of a few blocks (narrow scan lines; every few coded data sets hold a reference sample, and a
    // scoring chain without them stands on nothing): every bit starts a chain for EVERY count of blocks it could stand
    // at -- rsi times the chains; the one that scores stands on the true chain WITH the true count, which is the whole
    // entry: no walk to an RSI start.  State after three coded data sets in (pos, b_out); 0: none is convincing.
    auto find_anchor_phased = [&](uint64_t from, uint32_t &b_out) -> uint64_t {
        uint32_t best_sc = 0, best_b = 0;
        uint64_t best_at = 0;
        uint64_t pbeg = 0, pend = 0;
        if (fast) {
            (void)ws.ensure(from, WS::kPiece / 2u);
            pbeg = ws.base * 32u + ws.pc0;
            pend = pbeg + WS::kPiece;
        }
        for (uint32_t phase = 0; phase < c.rsi; phase++) {
            for (uint64_t q0 = from; q0 < from + maxbits + 64u; q0 += 64u) {
                uint64_t q = q0 + lane, at3 = 0;
                uint32_t sc = 0, prev = 0, b = phase, b3 = 0;
                bool ok = true;
                for (uint32_t i = 0; i < kLpSteps && ok; i++) {
                    const uint32_t ref = (pp && b == 0u) ? 1u : 0u;
                    uint32_t nz = 0, len = 0, id;
                    if (q >= pbeg && q < pend) {
                        const uint32_t rel = (uint32_t)(q - ws.base * 32u);
                        const uint32_t e = ws.entry(c, rel, ref);
                        len = e & 0xFFFu;
                        if (e & kNxtZero) nz = len - c.id_len - 1u - ref * c.bps;
                        id = ws.option(c, rel);
                    } else {
                        id = lp_id(s, c, q);
                    }
                    if (!len) len = tr_cds(s, c, q, ref, nz);
                    const uint32_t nb = len ? tr_blocks(c, nz, b) : 0u;
                    ok = nb != 0u;
                    if (i && near1(id, prev)) sc++;
                    prev = id;
                    if (i == 3u) {
                        at3 = q;
                        b3 = b;
                    }
                    q += len;
                    b += nb;
                    if (b >= c.rsi) b = 0u;
                }
                if (!ok) sc = 0;
                if (sc > best_sc) {
                    best_sc = sc;
                    best_at = at3;
                    best_b = b3;
                }
            }
        }
        uint32_t top = best_sc;
        for (int off = 32; off; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)top, off);
            top = o > top ? o : top;
        }
        const uint64_t who = __ballot(best_sc == top);
        const int src = (int)__builtin_ctzll(who);
        const uint64_t at = __shfl(best_at, src);
        b_out = (uint32_t)__shfl((int)best_b, src);
        return top >= kLpAccept ? at : 0ull;
    };
    // the coded data set that would begin at v, parsed by the lane's half of the wavefront out of the window (v and ref
    // are the same in the 32 lanes of a half; aec_stretch.h: coop_half); from device memory what that does not resolve
    uint64_t wbeg = 0, wend = 0;
    const uint32_t myref = pp ? (lane >> 5) : 0u;      // (the half of the wavefront that parses with a reference sample)
    auto cover = [&](uint64_t q, uint32_t span) {
        if (!fast) return;
        (void)ws.ensure_win(q, span);
        wbeg = ws.base * 32u;
        wend = wbeg + kSwWin * 32u;
    };
    auto parse = [&](uint64_t v, uint32_t ref, uint32_t &nz) -> uint32_t {
        uint32_t len = 0;
        nz = 0;
        if (fast) {
            const uint32_t rel = (v >= wbeg && v < wbeg + ws.wlim) ? (uint32_t)(v - wbeg) : 0x7FFFFF00u;
            const uint32_t e = ws.coop_half(c, rel, ref);
            len = e & 0xFFFu;
            if (e & kNxtZero) nz = len - c.id_len - 1u - ref * c.bps;
        }
#ifdef AEC_TUNING
        if (!len) atomicAdd(&g_lp_prof[7], 1ull);
#endif
        if (!len) len = tr_cds(s, c, v, ref, nz);
        return len;
    };
    auto opt = [&](uint64_t v) -> uint32_t {
        return (v >= wbeg && v + 64u < wend) ? ws.option(c, (uint32_t)(v - wbeg)) : lp_id(s, c, v);
    };
    // Two chains of up to kLpConfirm coded data sets from q0 on, one per half of the wavefront: the plain one (lanes 0..)
    // and the one whose first coded data set has a reference sample (lanes 32..), scored by the options that lie within
    // 2 of their predecessor's; where one does not fit, the coded data set in front is tried WITH a reference sample (a
    // true chain may pass an RSI start and still scores).  Verdict: 1 an RSI starts at q0 (the chain with the reference
    // sample is plausible, and more so than the plain one), 2 neither is plausible (the walk has lost the true chain),
    // 0 on with the plain one -- as soon as it is certain, or the plain chain is two ahead.
    auto confirm = [&](uint64_t q0, bool with_ref) -> uint32_t {
        const uint32_t n = kLpConfirm;
        uint32_t sc = 0, idp = opt(q0), so = 0, sr = 0;
        uint64_t v = q0;
        bool ok = true;
        for (uint32_t k = 0; k < n; k++) {
            uint32_t nz, len = parse(v, k == 0u ? myref : 0u, nz);
            ok = ok && len != 0u;
            uint32_t idk = ok ? opt(v + len) : ~0u;
            const bool doubt = ok && k && pp && !near2(idk, idp);
            if (__any(doubt)) {
                uint32_t nz1;
                const uint32_t l1 = parse(v, 1u, nz1);
                const uint32_t id1 = l1 ? opt(v + l1) : ~0u;
                if (doubt && l1 && near2(id1, idp)) {
                    len = l1;
                    idk = id1;
                }
            }
            if (ok) {
                v += len;
                sc += near2(idk, idp) ? 1u : 0u;
                idp = idk;
            }
            so = (uint32_t)__builtin_amdgcn_readlane((int)sc, 0);
            sr = with_ref ? (uint32_t)__builtin_amdgcn_readlane((int)sc, 32) : 0u;
            const uint32_t done = k + 1u;
            if (so >= sr + 2u && so >= 2u) return 0u;
            if (done - sr > n - kLpConfirmOk && (so >= kLpLost || done - so > n - kLpLost)) break;
        }
        if (sr >= kLpConfirmOk && sr > so) return 1u;
        return so < kLpLost ? 2u : 0u;
    };
    LkState guess{rstart, 0u, 0u};                      // (no guess: a wrong one, repaired or judged later)
    const unsigned long long lp_t0 = LP_NOW();
    if (c.rsi < kLpPhased) {                            // RSIs of a few blocks: the anchor is the entry, walked on exactly
        uint32_t b0 = 0;
        const uint64_t at = find_anchor_phased(A, b0);
        LP_ADD(0, LP_NOW() - lp_t0);
        LP_ADD(1, 1);
        if (at) {
            LkState x{at, b0, 0u};
            if (fast) {
                lk_walk_coop(ws, s, c, x, rstart, [](const LkState &) { return true; });
            } else {
                while (x.pos < rstart && !x.st) lk_step(s, c, x);
            }
            if (!x.st && x.pos >= rstart) guess = x;
            LP_ADD(8, 1);
            LP_ADD(9, 1);
        }
        LP_ADD(6, LP_NOW() - lp_t0);
        if (lane == 0) t.entry[r] = guess;
        return;
    }
    uint64_t q = find_anchor(A);
    LP_ADD(0, LP_NOW() - lp_t0);
    LP_ADD(1, 1);
    // ---- 2. along the chain WITHOUT reference samples, looking one coded data set ahead BOTH ways at every step (one half
    // of the wavefront parses without a reference sample, the other with one): as long as the option behind the plain parse fits and the one
    // behind the other does not, on it goes.  Anything else is decided by kLpConfirm coded data sets either way: an RSI
    // starts here if the chain WITH a reference sample is plausible and more so than the plain one (found); the plain
    // one if it is plausible (the data jumped); neither: the walk has lost the true chain and looks for it again.
    // The walk notes where it passes the region's start and counts the blocks from there: RSIs start every rsi blocks,
    // so the first RSI start behind the region's start gives the count at the entry.
    uint64_t cross = 0, S = 0;
    uint32_t steps = 0, nb_since = 0, reanch = 0;
    bool crossed = false, found = false, cnt_ok = true;
    // (two RSIs: an RSI start is passed unseen now and then, see below, and the next one serves as well.  A budget of an
    // RSI and a half made the slowest region, which is the kernel's duration, no faster to speak of and cost 1 GiB of the
    // sample file's shape 10 ms: its regions are 8 RSIs long, and so is every repair of a guess that failed.)
    const uint32_t most = 2u * c.rsi + 64u;
    while (q && steps++ < most) {
        if (!crossed && q >= rstart) {
            crossed = true;
            cross = q;
            nb_since = 0;
        }
        // (an RSI start is passed unseen where the plain parse of its first coded data set finds the true chain again at
        // once -- both ways look the same then, one in six on the sample file; the count of blocks mostly survives it,
        // and the next RSI start gives the same answer modulo rsi)
        cover(q, kLpSpan);
        uint32_t nzl;
        const uint32_t ll = parse(q, myref, nzl);
        const uint32_t id = (uint32_t)__builtin_amdgcn_readfirstlane((int)opt(q));
        const uint32_t idn = ll ? opt(q + ll) : ~0u;
        const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)ll, 0);
        const uint32_t nz0 = (uint32_t)__builtin_amdgcn_readlane((int)nzl, 0);
        const uint32_t l1 = pp ? (uint32_t)__builtin_amdgcn_readlane((int)ll, 32) : 0u;
        if (!l0) break;
        const bool nf = near1((uint32_t)__builtin_amdgcn_readlane((int)idn, 0), id);
        const bool ng = l1 && near1((uint32_t)__builtin_amdgcn_readlane((int)idn, 32), id);
        if (!(nf && !ng)) {
            const unsigned long long lp_t2 = LP_NOW();
            const uint32_t verdict = confirm(q, l1 != 0u);
            LP_ADD(2, LP_NOW() - lp_t2);
            LP_ADD(3, 1);
            if (verdict == 1u) {
                found = true;
                S = q;
                break;
            }
            if (verdict == 2u) {
                LP_ADD(14, 1);
                if (crossed || ++reanch > 2u) break;
                const unsigned long long lp_t4 = LP_NOW();
                q = find_anchor(q);
                LP_ADD(0, LP_NOW() - lp_t4);
                LP_ADD(1, 1);
                continue;
            }
        }
        q += l0;
        if (crossed) {
            if (nz0 == 5u) cnt_ok = false;              // (a rest-of-segment run: its blocks depend on the count)
            nb_since += nz0 ? (nz0 > 5u ? nz0 - 1u : nz0) : 1u;
        }
    }
    LP_ADD(4, LP_NOW() - lp_t0);
    LP_MAX(13, LP_NOW() - lp_t0);
    LP_ADD(5, steps);
    LP_ADD(8, found ? 1 : 0);
    LP_ADD(9, (found && !crossed) ? 1 : 0);
    // ---- 3. the entry
    if (found && !crossed) {                            // exact from the RSI start on to the region's start
        LkState x{S, 0u, 0u};
        if (fast) {
            lk_walk_coop(ws, s, c, x, rstart, [](const LkState &) { return true; });
        } else {
            while (x.pos < rstart && !x.st) lk_step(s, c, x);
        }
        if (!x.st && x.pos >= rstart) guess = x;
    } else if (found && cnt_ok) {                       // back from the RSI start: rsi less the blocks since the entry
        guess = LkState{cross, (c.rsi - nb_since % c.rsi) % c.rsi, 0u};
    }
    LP_ADD(6, LP_NOW() - lp_t0);
    LP_MAX(10, LP_NOW() - lp_t0);
    LP_MAX(11, steps);
    LP_MAX(12, reanch);
    if (lane == 0) t.entry[r] = guess;
}

// after the first walk: how many guesses did not hold?  Many: the options of this data say nothing; the stream is the
// trunk's (flags[3]; the kernels of this scheme return, the trunk's are no longer skipped -- flags[0] stays 0).
__global__ void __launch_bounds__(1024)
k_lock_judge(const LockTables t, const LkState *exit_last)
{
    if (t.skip_if && *t.skip_if) return;
    __shared__ uint32_t wrong;
    if (threadIdx.x == 0) wrong = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t r = 1u + threadIdx.x; r < t.nreg; r += blockDim.x) {
        const LkState prev = exit_last[r - 1u], en = t.entry[r];
        // (a walk from a wrong entry mostly ends refused -- a run of zero blocks that does not fit the RSI by its count)
        if (prev.st != 0u || prev.pos != en.pos || prev.b != en.b) mine++;
    }
    if (mine) atomicAdd(&wrong, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        t.flags[4] = wrong;                              // (statistics)
        if (t.nreg >= 8u && wrong * 4u > t.nreg) t.flags[3] = 1u;
    }
}

// "delivered" of the scheme in front is "delivered" here too (the serial walker behind looks at these flags alone)
__global__ void k_lock_merge(uint32_t *flags, const uint32_t *front)
{
    if (*front) flags[0] = 1u;
}

// What the parallel repair passes left: ONE wavefront goes through the regions in order and, wherever a region's entry is
// not the exit of the region in front, walks on from that exit -- region after region, entries, counts and exits
// rewritten -- until the walk stands on a stored entry again (from there on the stored walks hold).  A repair pass mends
// one region of a run of wrong guesses per launch; this mends a run at the speed of the walk, and only the runs.
__global__ void __launch_bounds__(64)
k_lock_fix_w(const Cfg c, const TrStream s, const LockTables t, LkState *ex)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t lk_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    if (t.flags[3]) return;
    WaveStream ws;
    ws.init(lk_lds, s, c);
    const bool fast = ws.usable();
    uint32_t r = 1;
    while (r < t.nreg) {
        const uint32_t q0 = r + lane;
        bool mis = false, dead = false;
        if (q0 < t.nreg) {
            const LkState prev = ex[q0 - 1u], mine = t.entry[q0];
            dead = prev.st != 0u;
            mis = !dead && (prev.pos != mine.pos || prev.b != mine.b);
        }
        const uint64_t mm = __ballot(mis), dd = __ballot(dead);
        const uint32_t fm = mm ? (uint32_t)__builtin_ctzll(mm) : 64u, fd = dd ? (uint32_t)__builtin_ctzll(dd) : 64u;
        if (fd < fm) return;                              // (the walk ended in front of it: nothing behind lives)
        if (fm == 64u) {
            r += 64u;
            continue;
        }
        uint32_t q = r + fm;
        LkState x = ex[q - 1u];
        for (;;) {
            if (lane == 0) t.entry[q] = x;
            const uint64_t rend = q + 1u == t.nreg ? ~0ull : t.lo + (uint64_t)(q + 1u) * t.region_bits;
            uint32_t n = 0;
            if (fast) {
                lk_walk_any(t.coop != 0u, ws, s, c, x, rend, [&](const LkState &) {
                    n++;
                    return true;
                });
            } else {
                while (x.pos < rend) {
                    n += x.b == 0u ? 1u : 0u;
                    lk_step(s, c, x);
                    if (x.st) break;
                }
            }
            if (lane == 0) {
                t.cnt[q] = n;
                ex[q] = x;
            }
            if (x.st) return;
            q++;
            if (q >= t.nreg) return;
            const LkState nx = t.entry[q];
            if (nx.pos == x.pos && nx.b == x.b) break;    // on a stored entry again
        }
        __threadfence_block();
        r = q;
    }
}

__global__ void __launch_bounds__(256)
k_lock_fill_w(const Cfg c, const TrStream s, const LockTables t, const LkState *exit_last, const uint32_t *__restrict__ words,
              uint64_t nwords, uint64_t *__restrict__ rsi_off, uint64_t max_rsi, DecResult *res, uint32_t tail_slot,
              uint64_t rsi_start_in, uint32_t start_block)
{
    if (t.skip_if && *t.skip_if) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t lk_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= t.nreg || t.flags[1] || t.flags[3] || r > t.flags[2]) return;
    const uint64_t rend = r + 1u == t.nreg ? ~0ull : t.lo + (uint64_t)(r + 1u) * t.region_bits;
    LkState x = t.entry[r];
    const uint64_t off = start_block ? 1u : 0u;          // (k_lock_fill: a walk that resumes inside an RSI)
    uint64_t idx = t.base[r] + off;
    if (off && r == 0u && max_rsi && lane == 0) rsi_off[0] = rsi_start_in;
    if (idx > max_rsi) return;
    auto start_in_front = [&]() -> uint64_t {
        for (uint32_t q = r; q-- > 0u;) {
            if (!t.cnt[q]) continue;
            LkState y = t.entry[q];
            const uint64_t qend = t.lo + (uint64_t)(q + 1u) * t.region_bits;
            uint64_t last = rsi_start_in;
            while (y.pos < qend) {
                if (y.b == 0u) last = y.pos;
                lk_step(s, c, y);
                if (y.st) break;
            }
            return last;
        }
        return rsi_start_in;
    };
    uint64_t cur = 0;
    bool met = false, clipped = false;
    WaveStream ws;
    ws.init(lk_lds + (size_t)wave * kSwWaveWords, s, c);
    auto at_start = [&](const LkState &y) {
        if (idx == max_rsi) {
            clipped = true;
            return false;
        }
        if (lane == 0) rsi_off[idx] = y.pos;
        cur = y.pos;
        met = true;
        idx++;
        return true;
    };
    if (ws.usable()) {
        lk_walk_any(t.coop != 0u, ws, s, c, x, rend, at_start);
    } else {
        while (x.pos < rend) {
            if (x.b == 0u && !at_start(x)) break;
            lk_step(s, c, x);
            if (x.st) break;
        }
    }
    if (!clipped && !x.st) return;
    if (lane != 0) return;
    if (clipped) {
        res->n_rsi = max_rsi;
        res->tail_blocks = 0;
        res->end_bit = x.pos;
        res->status = DEC_OK;
        res->pad = 0u;
        res->bad_rsi = ~0ull;
        if (tail_slot) rsi_off[max_rsi] = met ? cur : start_in_front();
        __threadfence();
        t.flags[0] = 1u;
        return;
    }
    if (x.st != 1u) return;
    // (st 1 is also what a unary part beyond the parser's reach reports -- a foreign encoder's coded data set of 12 800
    // bits: only within that reach of the end of the input does it mean that the input ended)
    if (s.end_bit - x.pos > kTrMaxScan) return;
    {
        BitReaderT<QuadFetch> br;
        br.init(QuadFetch{words, nwords}, s.end_bit, x.pos);
        uint32_t nblk = 1;
        if (skip_cds(br, c, (x.b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u, x.b, nblk) != DEC_NEED_INPUT) return;
    }
    if (idx == 0) return;
    res->n_rsi = idx - 1u;
    res->tail_blocks = x.b;
    res->end_bit = x.pos;
    res->status = DEC_OK;
    res->pad = 1u;
    res->bad_rsi = ~0ull;
    if (tail_slot) rsi_off[max_rsi] = met ? cur : start_in_front();
    __threadfence();
    t.flags[0] = 1u;
}

struct LockPlan {
    bool ok;
    uint32_t mode;            // 0: entries by 64 chains that agree (short RSIs); 1: by plausibility (k_lock_guess_p)
    uint32_t back;            // mode 1: how far in front of a region its guess begins (bits)
    uint32_t nreg, region_bits, lead;
    uint32_t gap;             // mode 0: bits between the starts of a guess's 64 chains
    uint32_t coop;            // the walks parse one coded data set at a time (long ones) instead of 64 bits at a time
    size_t o_entry, o_entry1, o_exit0, o_exit1, o_cnt, o_base, o_flags, bytes;
};

// the tables of either mode
static void lock_layout(LockPlan &p, uint64_t nreg)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    p.o_flags = o;  o = up(o + 64);
    p.o_entry = o;  o = up(o + nreg * sizeof(LkState));
    p.o_entry1 = o; o = up(o + nreg * sizeof(LkState));
    p.o_exit0 = o;  o = up(o + nreg * sizeof(LkState));
    p.o_exit1 = o;  o = up(o + nreg * sizeof(LkState));
    p.o_cnt = o;    o = up(o + nreg * 4);
    p.o_base = o;   o = up(o + (nreg + 1) * 8);
    p.bytes = o;
}

// Mode 1 (k_lock_guess_p): long coded data sets in RSIs that are short against the distance a chain needs to find the
// true one (2 cds^2 bits) -- rsi < 8 cds, where the trunk's hypotheses walk for RSIs on end; with the preprocessor (the
// guess looks for the coded data set that has a reference sample), fewer than 8 segments per RSI (no segment starts come
// out of this scheme), a stream of at least a few RSIs.
static LockPlan lock_plan_p(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint)
{
    LockPlan p{};
    if (!tune("AEC_IDX_LOCK_P", 1) || (c.flags & F_PAD_RSI) || !(c.flags & F_PREPROCESS) || !rsi_bits_hint) return p;
    if (index_incompressible(c, rsi_bits_hint)) return p;
    const uint64_t cds = rsi_bits_hint / c.rsi;
    // (RSIs of fewer than 16 blocks: a scoring chain of 16 coded data sets passes an RSI start or three -- there the
    // chains carry the count of blocks, one set of chains per count: k_lock_guess_p, find_anchor_phased)
    const uint64_t cds_min = (c.rsi <= kLockMaxRsi && c.id_len >= 4u) ? (uint64_t)tune("AEC_IDX_LOCK_P_CDS", 32)
                                                              : (uint64_t)tune("AEC_IDX_LOCK_P_CDS_LONG", 96);
    if (cds < cds_min || c.rsi >= 8 * cds || c.segs_per_rsi >= 8u || total_bits < 4 * rsi_bits_hint) return p;
    if (c.id_len + 1u + c.bps + c.bs * c.bps > (kSwLookWords - 2u) * 32u) return p;
    // regions of 1 .. 8 RSIs: large streams pay the guess (two RSIs walked per region) less often, small ones get
    // wavefronts to run
    // (RSIs of a few blocks: regions of 64 coded data sets at least -- the guess costs rsi x 18 x 16 parses)
    uint64_t region = (total_bits / 2048 + 1023) & ~1023ull;
    const uint64_t rmin = rsi_bits_hint > 64 * cds ? rsi_bits_hint : 64 * cds;
    if (region < rmin) region = rmin;
    if (region > 8 * rmin) region = 8 * rmin;
    region = (region + 1023) & ~1023ull;
    const uint64_t nreg = (total_bits + region - 1) / region;
    if (nreg > (1u << 24) || region > 0xFFFFFFFFull) return p;
    p.mode = 1;
    p.coop = cds >= (uint64_t)tune("AEC_IDX_LOCK_COOP_CDS", 96) ? 1u : 0u;
    p.nreg = (uint32_t)nreg;
    p.region_bits = (uint32_t)region;
    p.lead = 0;
    // (the scoring chains start in a stretch as long as the longest coded data set and stand three coded data sets further
    // on: in front of the region's start; the walk from there goes FORWARD to the next RSI start -- half an RSI on
    // average -- and the count at the entry comes from the blocks in between)
    p.back = 5u * (c.id_len + 1u + c.bps + c.bs * c.bps) + 128u;
    lock_layout(p, nreg);
    p.ok = true;
    return p;
}

// allow_p = false: the plan of the 64 agreeing chains alone (what the dispatch runs behind entries by plausibility that
// were judged wrong, before the trunk)
LockPlan lock_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint, uint32_t start_block, bool allow_p = true)
{
    LockPlan p{};
    // (without the preprocessor no coded data set holds a reference sample: nothing a chain could lock its count on)
    // (a walk that resumes inside an RSI numbers that RSI 0 and the first RSI start it meets 1: k_lock_fill)
    (void)start_block;
    // RSIs of at most 32 blocks -- or of any number of blocks that code to next to nothing: constant data, fill values
    // (an RSI of 128 zero blocks is two coded data sets, 26 bits; as many candidates per window as the tables of neither
    // kind have room for, and the serial walk took 64 ms for 64 MiB of a constant)
    const bool tiny = rsi_bits_hint != 0 && rsi_bits_hint <= tune("AEC_IDX_LOCK_TINY", 256u);
    if (!tune("AEC_IDX_LOCK", 1) || (c.flags & F_PAD_RSI) || !(c.flags & F_PREPROCESS)) return p;
    if (c.rsi > kLockMaxRsi && !tiny) return allow_p ? lock_plan_p(c, total_bits, rsi_bits_hint) : p;
    if (total_bits < (1u << 16)) return p;               // (a thousand coded data sets: the serial walker is as fast)
    uint64_t cds = rsi_bits_hint ? rsi_bits_hint / c.rsi : (uint64_t)(c.id_len + c.bs * c.bps) / 3;
    if (cds < 8) cds = 8;
    // lock distance ~ cds x rsi steps of cds bits; most of 64 chains are to be locked where the region begins
    // (measured with factors 4 / 2 / 1: 16-bit, rsi 16: 20 / 13 / 10 ms per 16 MiB; 32-bit, block 32, rsi 5: 67 / 38 / 24 ms per
    // 48 MiB; a guess that is wrong only costs a repair pass)
    // (round 5, a wavefront per region: swept again with factors 2 / 1 and regions of 1/1 .. 1/16 of the lead-in.  Factor 1
    // halves the guesses' cost and is faster where it locks -- 16 MiB of 16-bit data, rsi 32: 8.8 against 10.6 ms -- but
    // its guesses are wrong in longer runs, and it moves streams of long coded data sets from the trunk to this scheme
    // with lead-ins of megabits: 16 MiB of 24-bit data in blocks of 64, rsi 16: 380 ms instead of 6.  2 it stays.)
    uint64_t lead = (uint64_t)tune("AEC_IDX_LOCK_LEAD", 2) * cds * cds * c.rsi;
    const uint64_t lmin = tune("AEC_IDX_LOCK_LMIN", 4096);
    if (lead < lmin) lead = lmin;
    // (long coded data sets: a lead-in that is a good part of the stream is as good as a serial walk -- 1 MiB of 16-bit
    // data in blocks of 32, rsi 16: 17 ms, five times the reference on one core -- where the options of real data tell
    // the true chain from the others in a dozen coded data sets: mode 1)
    // (and in RSIs of a few blocks whatever the size of the stream: 16 MiB of 24-bit data in blocks of 64 with rsi 1 took
    // 122 ms with lead-ins of 1.3 Mbit per region)
    // (and with 16 .. 32 blocks per RSI whatever the length of the coded data sets, where the options have four bits and
    // more -- with three, 8-bit data, half of the guesses are wrong: 16 MiB of 16-bit data in blocks of 16, rsi 32: 11.3
    // -> 3.2 ms, rsi 16: 6.4 -> 2.9; a walk to the next RSI start is 32 coded data sets at most)
    if (allow_p && ((cds >= 96 && (lead * 8 > total_bits || c.rsi < kLpPhased)) ||
                    (c.rsi >= kLpPhased && c.id_len >= 4u && cds >= (uint64_t)tune("AEC_IDX_LOCK_P_CDS", 32)))) {
        const LockPlan q = lock_plan_p(c, total_bits, rsi_bits_hint);
        if (q.ok) return q;
    }
    // (long coded data sets: too far to lock, unless the stream is long enough for a number of such regions)
    if (lead > (1u << 24) || (lead > (1u << 22) && total_bits < 4 * lead))
        return allow_p ? lock_plan_p(c, total_bits, rsi_bits_hint) : p;
    // (small streams: short regions -- the pass is as long as one lane's walk of a region, three times over)
    const uint64_t rmin = total_bits < (1u << 22) ? tune("AEC_IDX_LOCK_RMIN", 1024) : 16384;
    // (a region per wavefront now: regions a fraction of the lead-in, so that the walks -- one behind the other: count,
    // repairs, fill -- are short and the chip has wavefronts to run; the guesses cost the lead-in per region either way)
    const uint64_t rdiv = tune("AEC_IDX_LOCK_DIV", 16);
    uint64_t region = lead / (rdiv ? rdiv : 1) < rmin ? rmin : lead / (rdiv ? rdiv : 1);
    region = (region + 1023) & ~1023ull;
    const uint64_t nreg = (total_bits + region - 1) / region;
    if (nreg > (1u << 24)) return p;
    p.mode = 0;
    p.back = 0;
    p.nreg = (uint32_t)nreg;
    p.region_bits = (uint32_t)region;
    p.lead = (uint32_t)lead;
    // (the chains start a few RSIs apart in all -- 37 bits each were 2.4 kbit, more than the piece of the stream the
    // wavefront's tables cover: with coded data sets of 24 bits in RSIs of 4 most chains waited most of the time, and
    // the guesses of a 64 KiB chunk with scan lines of 32 pixels took 0.28 ms of its 0.59)
    uint64_t gap = cds * (c.rsi < 4u ? 4u : c.rsi) / 16;
    p.gap = (uint32_t)(gap < 3 ? 3 : gap > 37 ? 37 : gap);
    lock_layout(p, nreg);
    p.ok = true;
    return p;
}

// What runs behind guesses by plausibility that were judged wrong, before the trunk: the 64 agreeing chains -- unless
// their lead-in is a good part of the stream and RSIs have 16 blocks and more (long coded data sets: lead-ins of
// megabits, the trunk is faster: 1 MiB of 16-bit data in blocks of 32, rsi 16: 17 against 7 ms; with fewer blocks the
// trunk walks every RSI by itself -- 48 MiB with rsi 1: 1.5 s).
static LockPlan lock_plan_alt(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint, uint32_t start_block)
{
    if (c.rsi > kLockMaxRsi) return LockPlan{};
    const LockPlan l0 = lock_plan(c, total_bits, rsi_bits_hint, start_block, false);
    if (l0.ok && c.rsi >= kLpPhased && (uint64_t)l0.lead * 8u > total_bits) return LockPlan{};
    return l0;
}

void launch_index_locked(const Cfg &c, const LockPlan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                         uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                         uint8_t *base, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot,
                         bool serial_fallback = true, const uint32_t *skip_if = nullptr)
{
    const TrStream s{words, nwords, end_bit};
    LockTables t{};
    t.flags = reinterpret_cast<uint32_t *>(base + p.o_flags);
    t.entry = reinterpret_cast<LkState *>(base + p.o_entry);
    t.exit0 = reinterpret_cast<LkState *>(base + p.o_exit0);
    t.exit1 = reinterpret_cast<LkState *>(base + p.o_exit1);
    t.cnt = reinterpret_cast<uint32_t *>(base + p.o_cnt);
    t.base = reinterpret_cast<uint64_t *>(base + p.o_base);
    t.nreg = p.nreg;
    t.region_bits = p.region_bits;
    t.lead = p.lead;
    t.gap = p.gap ? p.gap : 37u;
    t.coop = p.coop;
    t.mode = p.mode;
    t.skip_if = skip_if;
    t.lo = start_bit;
    (void)hipMemsetAsync(t.flags, 0, 64, st);
    // a wavefront per region (k_lock_*_w) -- unless the parameters are beyond its tables' look-ahead
    const bool wave = p.mode == 1u ||
                      (tune("AEC_IDX_LOCK_WAVE", 1) != 0 && c.id_len + 1u + c.bps + c.bs * c.bps <= (kSwLookWords - 2u) * 32u);
    const uint32_t wpw = 4;                                             // wavefronts per workgroup
    const size_t wlds = (size_t)wpw * kSwWaveWords * 4;
    if (wave) {
        LkState *ex[2] = {t.exit0, t.exit1};
        if (p.nreg > 1 && p.mode == 1u)
            hipLaunchKernelGGL(k_lock_guess_p, dim3(p.nreg - 1), dim3(64), (size_t)sw_wave_words(kLpPiece) * 4, st, c, s, t, p.back);
        else if (p.nreg > 1)
            hipLaunchKernelGGL(k_lock_guess_w, dim3((p.nreg - 1 + wpw - 1) / wpw), dim3(64 * wpw), wlds, st, c, s, t);
        const uint32_t wg = (p.nreg + wpw - 1) / wpw;
        LkState *en[2] = {t.entry, reinterpret_cast<LkState *>(base + p.o_entry1)};
        hipLaunchKernelGGL(k_lock_walk_w, dim3(wg), dim3(64 * wpw), wlds, st, c, s, t, (const LkState *)nullptr, ex[0], en[0], 0u,
                           start_bit, start_block);
        if (p.mode == 1u) hipLaunchKernelGGL(k_lock_judge, dim3(1), dim3(1024), 0, st, t, (const LkState *)ex[0]);
#ifdef AEC_TUNING
        if (tune_set("AEC_IDX_DUMP")) {                    // (diagnostics: the guesses against the exits in front)
            (void)hipStreamSynchronize(st);
            std::vector<LkState> en0(p.nreg), ex0(p.nreg);
            (void)hipMemcpy(en0.data(), en[0], p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
            (void)hipMemcpy(ex0.data(), ex[0], p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
            uint32_t shown = 0;
            for (uint32_t r = 1; r < p.nreg && shown < 24; r++) {
                const bool mis = ex0[r - 1].pos != en0[r].pos || ex0[r - 1].b != en0[r].b;
                if (!mis && r > 6) continue;
                shown++;
                fprintf(stderr, "  region %u (from bit %llu): guess (%llu, %u) | exit in front (%llu, %u, st %u)%s\n", r,
                        (unsigned long long)(start_bit + (uint64_t)r * p.region_bits), (unsigned long long)en0[r].pos, en0[r].b,
                        (unsigned long long)ex0[r - 1].pos, ex0[r - 1].b, ex0[r - 1].st, mis ? "  <-- differ" : "");
            }
        }
#endif
        uint32_t cur = 0;
        // a few parallel repair passes (each mends the first region of every run of regions in doubt; an idle one is a
        // launch of 5 us, and 48 of them were a quarter of a millisecond on a 64 KiB chunk), then one wavefront mends
        // what is left run by run at the speed of the walk
        // (a small chunk: four passes -- one costs as much as the walk of a region, which is what the wavefront behind
        // them takes per region it mends; sixteen were 0.19 of the 0.59 ms of a 64 KiB chunk.  Not for 16 MiB: 16-bit
        // data with rsi 32 went from 11 to 33 ms with four.)
        const uint32_t passes = tune("AEC_IDX_LOCK_PASSES", p.mode == 1u ? 8u : p.nreg <= 256u ? 4u : 16u);
        for (uint32_t k = 0; k < passes; k++) {
            t.entry = en[cur];
            hipLaunchKernelGGL(k_lock_walk_w, dim3(wg), dim3(64 * wpw), wlds, st, c, s, t, (const LkState *)ex[cur], ex[cur ^ 1u],
                               en[cur ^ 1u], 1u, start_bit, start_block);
            cur ^= 1u;
        }
        t.entry = en[cur];
        hipLaunchKernelGGL(k_lock_fix_w, dim3(1), dim3(64), (size_t)kSwWaveWords * 4, st, c, s, t, ex[cur]);
        hipLaunchKernelGGL(k_lock_scan, dim3(1), dim3(1024), 0, st, t, (const LkState *)ex[cur]);
        hipLaunchKernelGGL(k_lock_fill_w, dim3(wg), dim3(64 * wpw), wlds, st, c, s, t, (const LkState *)ex[cur], words, nwords,
                           d_rsi_off, max_rsi, d_res, tail_slot, rsi_start, start_block);
#ifdef AEC_TUNING
        if (tune_set("AEC_IDX_STATS")) {                   // (diagnostics: synchronises)
            (void)hipStreamSynchronize(st);
            uint32_t fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpy(fl, t.flags, 32, hipMemcpyDeviceToHost);
            std::vector<LkState> en(p.nreg), exs(p.nreg);
            (void)hipMemcpy(en.data(), t.entry, p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
            (void)hipMemcpy(exs.data(), ex[cur], p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
            uint32_t mism = 0;
            for (uint32_t r = 1; r < p.nreg; r++) mism += exs[r - 1].st == 0 && (exs[r - 1].pos != en[r].pos || exs[r - 1].b != en[r].b);
            if (p.mode == 1u) {
                unsigned long long h[16];
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lp_prof), sizeof(h));
                const double n = p.nreg > 1 ? p.nreg - 1 : 1, us = 1e-2;   // (100 MHz shader clock counter)
                fprintf(stderr, "guesses, per region: %.1f us (anchors %.2f: %.1f us; walk %.0f steps with %.1f confirmations: %.1f us; "
                        "all but the exact walk %.1f us), %.1f parses from device memory; RSI start found %.0f%%, in front of the "
                        "region %.0f%% | the slowest: %.1f us (%.1f without the exact walk), most steps %llu, most new anchors %llu; chain lost %llu times\n",
                        h[6] / n * us, h[1] / n, h[0] / n * us, h[5] / n, h[3] / n, h[2] / n * us, h[4] / n * us,
                        h[7] / n, 100.0 * h[8] / n, 100.0 * h[9] / n, h[10] * us, h[13] * us, h[11], h[12], h[14]);
                unsigned long long z[16] = {0};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lp_prof), z, sizeof(z));
            }
            fprintf(stderr, "locked chains (mode %u): %u regions of %u bits | delivered %u, inconsistent %u, walk ended in region %u, "
                    "abandoned %u | guesses that did not hold after the first walk: %u; entries that are not the exit in front "
                    "after repairs: %u\n", p.mode, p.nreg, p.region_bits, fl[0], fl[1], fl[2], fl[3], fl[4], mism);
        }
#endif
        if (skip_if) hipLaunchKernelGGL(k_lock_merge, dim3(1), dim3(1), 0, st, t.flags, skip_if);
        if (!serial_fallback) return;                      // (the caller enqueues the trunk behind, skipped if this delivered)
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res,
                           (const uint64_t *)nullptr, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u, 1u, start_block, rsi_start,
                           tail_slot, TwTables{}, (ChunkEntry *)nullptr, SparseTables{}, (uint32_t *)nullptr, (uint64_t)0,
                           (const uint32_t *)t.flags);
        return;
    }
    if (p.nreg > 1)
        hipLaunchKernelGGL(k_lock_guess, dim3(p.nreg - 1 < 65536u ? p.nreg - 1 : 65536u), dim3(64), 0, st, c, s, t);
    const uint32_t wgrid = (p.nreg + 63) / 64;
    LkState *ex[2] = {t.exit0, t.exit1};
    hipLaunchKernelGGL(k_lock_walk, dim3(wgrid), dim3(64), 0, st, c, s, t, (const LkState *)nullptr, ex[0], 0u, start_bit,
                       start_block);
    uint32_t cur = 0;
    // (a stretch of k regions with wrong guesses -- incompressible data, where a chain needs far longer to lock -- takes
    // k passes; a pass with nothing to repair is a few microseconds)
    const uint32_t passes = tune("AEC_IDX_LOCK_PASSES", 48);
    for (uint32_t k = 0; k < passes; k++) {
        hipLaunchKernelGGL(k_lock_walk, dim3(wgrid), dim3(64), 0, st, c, s, t, (const LkState *)ex[cur], ex[cur ^ 1u], 1u,
                           start_bit, start_block);
        cur ^= 1u;
    }
    hipLaunchKernelGGL(k_lock_scan, dim3(1), dim3(1024), 0, st, t, (const LkState *)ex[cur]);
    hipLaunchKernelGGL(k_lock_fill, dim3(wgrid), dim3(64), 0, st, c, s, t, (const LkState *)ex[cur], words, nwords, d_rsi_off,
                       max_rsi, d_res, tail_slot, rsi_start, start_block);
#ifdef AEC_TUNING
    if (tune_set("AEC_IDX_STATS")) {                       // (diagnostics: synchronises)
        (void)hipStreamSynchronize(st);
        uint32_t fl[4] = {0, 0, 0, 0};
        (void)hipMemcpy(fl, t.flags, 16, hipMemcpyDeviceToHost);
        std::vector<LkState> en(p.nreg), exs(p.nreg);
        (void)hipMemcpy(en.data(), t.entry, p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
        (void)hipMemcpy(exs.data(), ex[cur], p.nreg * sizeof(LkState), hipMemcpyDeviceToHost);
        uint32_t mism = 0, ended = 0, refused = 0;
        for (uint32_t r = 1; r < p.nreg; r++) mism += exs[r - 1].st == 0 && (exs[r - 1].pos != en[r].pos || exs[r - 1].b != en[r].b);
        for (uint32_t r = 0; r < p.nreg; r++) {
            ended += exs[r].st == 1;
            refused += exs[r].st == 2;
        }
        fprintf(stderr, "locked chains: %u regions of %u bits, lead %u | delivered %u, inconsistent %u, walk ended in region %u | "
                "entries that are not the exit in front: %u, regions ended %u, refused %u\n", p.nreg, p.region_bits, p.lead, fl[0],
                fl[1], fl[2], mism, ended, refused);
    }
#endif
    if (skip_if) hipLaunchKernelGGL(k_lock_merge, dim3(1), dim3(1), 0, st, t.flags, skip_if);
    if (!serial_fallback) return;
    // whatever was not delivered: the serial walker, which returns at once otherwise
    hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res,
                       (const uint64_t *)nullptr, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u, 1u, start_block, rsi_start,
                       tail_slot, TwTables{}, (ChunkEntry *)nullptr, SparseTables{}, (uint32_t *)nullptr, (uint64_t)0,
                       (const uint32_t *)t.flags);
}

// ---- small streams: EVERY bit parsed, the chain of RSI starts by pointer doubling ---------------------------------
// A chunk of an HDF5 dataset is a call of its own (reference src/sz_compat.c:239): 64 KiB of pixels, a stream of 200
// kbit.  Every scheme above pays lead-ins, repair passes or tables laid out for gigabits: 0.3 .. 1.7 ms for such a
// chunk with RSIs of 1 .. 32 blocks, where the reference takes 0.1 .. 0.5 on one core.  A stream that small can be
// parsed at EVERY bit by brute force -- 400 k parses are a few microseconds of this chip:
//   1. k_small_parse: the coded data set that would begin at every bit, without and with a reference sample
//      (nxt[] format, exact or 0);
//   2. k_small_rsi:   from every bit one whole RSI (its first coded data set with a reference sample, the RSI's own
//      bookkeeping of zero-block runs): where the NEXT RSI would start, J[q];
//   3. k_small_double, log4(RSIs) times: the RSI starts 4^k .. 4^(k+1) - 1 from the first 4^k through J^(4^k) applied
//      once, twice, three times, and its fourth power for the next round;
//   4. k_small_finish: offsets, the walker's record, the trailing incomplete RSI walked by one lane -- anything out
//      of the ordinary (a coded data set that does not parse, a run that does not fit) leaves the stream to the
//      serial walker behind, as everywhere.
// Works for any parameter set (no preprocessor, any rsi up to kSmMaxRsi), needs no guesses; 8 bytes of workspace per bit.
constexpr uint64_t kSmMaxBits = 1ull << 24;      // 2 MiB of stream
constexpr uint32_t kSmMaxRsi = 64;               // (step 2 is rsi dependent loads per bit)
constexpr uint32_t kSmGrid = 4096;               // most workgroups of 256 per launch (16 per CU); the kernels stride

struct SmallPlan {
    bool ok;
    uint32_t nbits, scap, levels;     // bits of a piece (the whole stream if it is one), RSI starts of a piece, rounds
    uint32_t npieces;
    uint32_t hops;                    // RSIs of more than 16 blocks: step 2 goes through a table of 8 coded data sets per entry
                                      // (2: and, for more than 256 blocks, through one of 8 such hops per entry)
    size_t o_e0, o_e1, o_ja, o_jb, o_s, o_h, o_h2, o_flags, bytes;
};
// Streams beyond kSmMaxBits go PIECE BY PIECE (without the preprocessor only: the other schemes' chains have nothing to
// lock a count on there and such streams went over the trunk or to the serial walker -- 16 MiB of 16-bit data 31 .. 41
// ms, three times the reference on one core): a piece ends on the start of the RSI it does not hold whole, the next
// begins there; the cursor lives on the device, the launches of all pieces are enqueued at once.
struct SmCursor {
    uint64_t bit, idx, last_start;    // where the next piece begins, RSI starts delivered so far, the last one of them
    uint32_t stop, pad;               // no further piece (delivered, or left to the serial walker)
};

// forced: the plan as the FALLBACK behind a scheme that may give a stream up (the chains by plausibility whose entries
// were judged wrong, the window tables that resolve too few RSIs: launch_index) -- the limits of the scheme itself
// without the rules about where another scheme is faster, since that scheme has just failed.
static SmallPlan small_plan(const Cfg &c, uint64_t total_bits, uint64_t max_rsi, uint32_t start_block, uint64_t rsi_bits_hint,
                            bool forced = false)
{
    SmallPlan p{};
    if (!tune("AEC_IDX_SMALL", 1) || !max_rsi || total_bits < 64) return p;
    // (round 6: AEC_PAD_RSI -- the next RSI begins on a byte: sm_rsi rounds up -- and walks that resume inside an RSI --
    // the first walk begins with start_block blocks done -- are this scheme's too; the latter in one piece)
    if (start_block && total_bits > kSmMaxBits) return p;
    const bool pp = c.flags & F_PREPROCESS;
    // Where the scheme runs, by measurement (tests/bench_short_rsi.py, --edges; tests/fuzz_index_gpu.py --time):
    //  * more than one piece: only without the preprocessor (no reference samples: the other schemes' chains have nothing
    //    to lock a count on) -- and with RSIs of more than 64 blocks only for samples of more than 8 bits (16 MiB with rsi
    //    128 / 256: 16-bit 34 / 40 -> 12 / 14 ms, the reference 13; 8-bit 8.6 / 8.4 over the trunk against 11.8 / 13.7 here);
    //    -- and, with it, streams of up to 4 MiB whose entries would be guessed by plausibility: the guesses alone take as
    //    long as this scheme's pass (4 MiB of 16-bit data, rsi 256: 4.2 of 7.2 ms when they are judged wrong, 5.4 against
    //    3.0 when they hold; 1 MiB of 8-bit data in blocks of 32, rsi 64: 1.5 against 0.45)
    //    (RSIs of more than 32 blocks; with 16 .. 32 the guesses are cheap: 16 MiB of 16-bit data, rsi 32: 2.9 against 3.5 ms)
    const bool by_plausibility = pp && c.rsi > kLockMaxRsi && total_bits <= (uint64_t)tune("AEC_IDX_SMALL_PP_BITS", 1u << 25) &&
                                 lock_plan(c, total_bits, rsi_bits_hint, start_block).mode == 1u;
    const bool pp_pieces = pp && (forced || by_plausibility);
    if (total_bits > kSmMaxBits && ((pp && !pp_pieces) || (c.rsi > kSmMaxRsi && c.bps <= 8u) || !tune("AEC_IDX_SMALL_PIECES", 1) ||
                                    total_bits >= (1ull << 40)))
        return p;
    //  * RSIs of more than 256 blocks (hops of hops): one piece;
    if (c.rsi > 256u && (total_bits > kSmMaxBits || !tune("AEC_IDX_SMALL_HOP2", 1))) return p;
    //  * with the preprocessor, where the window tables serve the stream: RSIs of up to 44 blocks always (the tables do
    //    not resolve them); 45 .. 64 up to 512 KiB of stream (5 MiB of 8-bit data with rsi 64: 1.1 against 1.7 ms); more
    //    than 64 up to 128 KiB (the 8-bit SZIP shape, rsi 128: 64 KiB 0.24 -> 0.19 ms, 1 MiB 0.39 against 0.49).  Where
    //    the tables do not serve it -- long coded data sets, high entropy -- the stream would go over the trunk, whose
    //    dozen launches cost a 1 MiB stream 4 .. 9 ms: here it stays.
    if (pp && c.rsi > kLockMaxRsi && !forced && !by_plausibility) {
        const uint64_t most_bits = c.rsi <= (uint32_t)tune("AEC_IDX_SMALL_RSI", kSmMaxRsi) ? (1u << 22) : (uint64_t)tune("AEC_IDX_SMALL_RSI_BITS", 1u << 20);
        if (total_bits > most_bits && sparse2_plan(c, total_bits, rsi_bits_hint).ok) return p;
    }
    const uint64_t piece = total_bits < kSmMaxBits ? total_bits : kSmMaxBits;
    // (what an ENCODER makes of an RSI at most; a piece holds a few of them or the scheme is not for this stream)
    const uint64_t worst = (uint64_t)c.rsi * (c.id_len + (uint64_t)c.bs * c.bps) + c.bps + 8;
    if (total_bits > kSmMaxBits && worst * 4 > piece) return p;
    const uint64_t min_rsi_bits = (uint64_t)c.segs_per_rsi * (c.id_len + 2u) + (pp ? c.bps : 0u);
    uint64_t most = piece / min_rsi_bits + 2;            // RSI starts a piece can hold ...
    if (most > max_rsi + 1) most = max_rsi + 1;          // ... and the caller asks for (+ the one that clips)
    uint32_t levels = 0;                                 // rounds of the doubling, each a factor of FOUR (a launch is 5 us)
    while ((1ull << (2u * levels)) < most) levels++;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    p.nbits = (uint32_t)piece;
    p.scap = 1u << (2u * levels);
    p.levels = levels;
    p.npieces = total_bits <= kSmMaxBits ? 1u : (uint32_t)(total_bits / (piece - worst) + 2);
    size_t o = 0;
    p.o_flags = o; o = up(o + 64 + sizeof(SmCursor));
    p.o_e0 = o;    o = up(o + ((size_t)p.nbits + 1) * 2);
    p.o_e1 = o;    o = up(o + ((size_t)p.nbits + 1) * 2);
    p.o_jb = p.o_e0;                                     // (the second table of the doubling: over the parses, done with by then)
    p.o_ja = o;    o = up(o + ((size_t)p.nbits + 1) * 4);
    p.o_s = o;     o = up(o + (size_t)p.scap * 4);
    p.hops = c.rsi > 256u ? 2u : (c.rsi > (uint32_t)tune("AEC_IDX_SMALL_HOP_RSI", 16) ? 1u : 0u);
    p.o_h = o;
    if (p.hops) o = up(o + ((size_t)p.nbits + 1) * 4);
    p.o_h2 = o;
    if (p.hops > 1u) o = up(o + ((size_t)p.nbits + 1) * 4);
    p.bytes = o;
    p.ok = true;
    return p;
}

// the piece at hand: [start, start + nbits); false: nothing left to do
__device__ __forceinline__ bool sm_piece(const SmCursor *cur, uint64_t end_bit, uint32_t piece_bits, uint64_t &start, uint32_t &nbits)
{
    if (cur->stop) return false;
    start = cur->bit;
    const uint64_t left = end_bit > start ? end_bit - start : 0u;
    nbits = (uint32_t)(left < piece_bits ? left : piece_bits);
    return true;
}

__global__ void __launch_bounds__(256)
k_small_parse(const Cfg c, const TrStream s, const SmCursor *cur, uint32_t piece_bits, uint16_t *__restrict__ e0,
              uint16_t *__restrict__ e1)
{
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, s.end_bit, piece_bits, start_bit, nbits)) return;
    // (the grids of this scheme's kernels are capped and the kernels stride: a piece that does not run -- the cursor says
    // stop, or the scheme in front has delivered -- costs a launch of kSmGrid workgroups that return, not of 65 536)
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q <= nbits; q += gridDim.x * blockDim.x) {
        uint16_t a = 0, b = 0;
        if (q < nbits) sm_parse(s, c, start_bit + q, a, b);
        e0[q] = a;
        e1[q] = b;
    }
}

// Step 2 walks a whole RSI from every bit: rsi dependent reads.  For RSIs of more than 16 blocks a table of HOPS first:
// from every bit up to 8 coded data sets without reference samples -- bits (15) and blocks (9 bits above) covered; a run
// of zero blocks to the end of its segment (its length depends on where in the RSI it stands) ends a hop in front of it.
// The walk then takes a hop wherever its blocks still fit the RSI and single coded data sets elsewhere (the first,
// with its reference sample; rest-of-segment runs; the last few): rsi / 8 + a dozen reads instead of rsi.
__global__ void __launch_bounds__(256)
k_small_hop(const Cfg c, const SmCursor *cur, uint64_t end_bit, uint32_t piece_bits, const uint16_t *__restrict__ e0,
            uint32_t *__restrict__ hop)
{
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, end_bit, piece_bits, start_bit, nbits)) return;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q <= nbits; q += gridDim.x * blockDim.x)
        hop[q] = sm_hop(c, [&](uint32_t at) { return (uint32_t)e0[at]; }, q, nbits);
}

__global__ void __launch_bounds__(256)
k_small_hop2(const SmCursor *cur, uint64_t end_bit, uint32_t piece_bits, const uint32_t *__restrict__ hop, uint32_t *__restrict__ hop2)
{
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, end_bit, piece_bits, start_bit, nbits)) return;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q <= nbits; q += gridDim.x * blockDim.x)
        hop2[q] = sm_hop2([&](uint32_t at) { return hop[at]; }, q, nbits);
}

// (the walks of a workgroup begin at consecutive bits and stay within an RSI's length of them: the table they read at
// every step -- the parses, or the hops -- is staged in LDS for that stretch; what leaves the stretch reads memory)
constexpr uint32_t kSmRsiWg = 1024, kSmRsiSpan = 8192;

__global__ void __launch_bounds__(kSmRsiWg)
k_small_rsi(const Cfg c, const SmCursor *cur, uint64_t end_bit, uint32_t piece_bits, const uint16_t *__restrict__ e0,
            const uint16_t *__restrict__ e1, const uint32_t *__restrict__ hop, const uint32_t *__restrict__ hop2,
            uint32_t *__restrict__ j, uint32_t *__restrict__ sidx, uint32_t scap, uint32_t start_block)
{
    __shared__ uint32_t lds[kSmRsiWg + kSmRsiSpan];
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, end_bit, piece_bits, start_bit, nbits)) return;
    const uint32_t pad = (c.flags & F_PAD_RSI) ? 1u + (uint32_t)(start_bit & 7u) : 0u;
    const uint32_t most = nbits + 1u > scap ? nbits + 1u : scap;
    for (uint32_t w0 = blockIdx.x * blockDim.x; w0 < most; w0 += gridDim.x * blockDim.x) {
        const uint32_t q = w0 + threadIdx.x;
        if (q < scap && (q || !start_block)) sidx[q] = q ? kSmNone : 0u;
        const uint32_t wn = w0 > nbits ? 0u : (nbits + 1u - w0 < kSmRsiWg + kSmRsiSpan ? nbits + 1u - w0 : kSmRsiWg + kSmRsiSpan);
        // (staged: the hops, or both parses of a position in one word)
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < wn; i += blockDim.x)
            lds[i] = hop ? hop[w0 + i] : ((uint32_t)e0[w0 + i] | ((uint32_t)e1[w0 + i] << 16));
        __syncthreads();
        if (q > nbits) continue;
        // (what is staged comes out of LDS: the hops, or both parses of a position in one word)
        auto r0 = [&](uint32_t at) { return (!hop && at - w0 < wn) ? (lds[at - w0] & 0xFFFFu) : (uint32_t)e0[at]; };
        auto r1 = [&](uint32_t at) { return (!hop && at - w0 < wn) ? (lds[at - w0] >> 16) : (uint32_t)e1[at]; };
        auto rh = [&](uint32_t at) { return at - w0 < wn ? lds[at - w0] : hop[at]; };
        auto rh2 = [&](uint32_t at) { return hop2[at]; };
        j[q] = sm_rsi(c, r0, r1, rh, hop != nullptr, rh2, hop2 != nullptr, q, nbits, 0u, pad);
        // a walk that resumes inside an RSI: the first RSI start of the chain is where THAT RSI ends
        if (q == 0u && start_block) sidx[0] = sm_rsi(c, r0, r1, rh, hop != nullptr, rh2, hop2 != nullptr, 0u, nbits, start_block, pad);
    }
}

// round k, with quarter = 4^k known RSI starts: sidx[m * quarter + i] = j^m[sidx[i]] for m = 1 .. 3, and jn = j^4
__global__ void __launch_bounds__(256)
k_small_double(const SmCursor *cur, uint64_t end_bit, uint32_t piece_bits, const uint32_t *__restrict__ j, uint32_t *__restrict__ jn,
               uint32_t *__restrict__ sidx, uint32_t quarter, uint32_t scap, uint32_t last)
{
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, end_bit, piece_bits, start_bit, nbits)) return;
    const uint32_t most = last ? quarter : (nbits + 1u > quarter ? nbits + 1u : quarter);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < most; q += gridDim.x * blockDim.x) {
        if (q < quarter) sm_double_starts(j, sidx, q, quarter, scap);
        if (!last && q <= nbits) jn[q] = sm_double_table(j, q);
    }
}

__global__ void __launch_bounds__(1024)
k_small_finish(const Cfg c, const TrStream s, SmCursor *cur, uint32_t piece_bits, const uint32_t *__restrict__ sidx, uint32_t scap,
               const uint32_t *__restrict__ words, uint64_t nwords, uint64_t *__restrict__ rsi_off, uint64_t max_rsi,
               DecResult *res, uint32_t tail_slot, uint32_t *flags, uint32_t start_block)
{
    __shared__ uint32_t first_none;
    uint64_t start_bit;
    uint32_t nbits;
    if (!sm_piece(cur, s.end_bit, piece_bits, start_bit, nbits)) return;
    const uint64_t base = cur->idx, before = cur->last_start;
    if (threadIdx.x == 0) first_none = scap;
    __syncthreads();
    uint32_t mine = scap;
    for (uint32_t i = threadIdx.x; i < scap; i += blockDim.x)
        if (sidx[i] == kSmNone) {
            mine = i;
            break;
        }
    if (mine < scap) atomicMin(&first_none, mine);
    __syncthreads();
    const uint32_t m = first_none;                      // RSI starts of the chain in this piece (the first is its start)
    const bool last_piece = start_bit + nbits >= s.end_bit;
    const bool clips = base + m > max_rsi;              // the start of RSI max_rsi lies in this piece
    // the starts this piece delivers: all it has -- but the last one is the next piece's first, if there is a next piece
    uint64_t nout = (last_piece || clips) ? m : m - 1u;
    if (base + nout > max_rsi) nout = max_rsi - base;
    for (uint64_t i = threadIdx.x; i < nout; i += blockDim.x) rsi_off[base + i] = start_bit + sidx[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (clips) {                                        // the caller's bound: ends on the start of RSI max_rsi
        const uint32_t l = (uint32_t)(max_rsi - base);
        res->n_rsi = max_rsi;
        res->tail_blocks = 0;
        res->end_bit = start_bit + sidx[l];
        res->status = DEC_OK;
        res->pad = 0u;
        res->bad_rsi = ~0ull;
        if (tail_slot) rsi_off[max_rsi] = l ? start_bit + sidx[l - 1u] : before;
        cur->stop = 1u;
        __threadfence();
        flags[0] = 1u;
        return;
    }
    if (!last_piece) {
        if (m < 2u) {                                   // (an RSI that a piece does not hold: the serial walker's)
            cur->stop = 1u;
            return;
        }
        cur->bit = start_bit + sidx[m - 1u];
        cur->idx = base + m - 1u;
        cur->last_start = start_bit + sidx[m - 2u];
        return;
    }
    cur->stop = 1u;
    // the RSI that does not end inside the input, walked as the serial walker walks it (skip_cds: the parse that also
    // sees second-extension codes beyond the table); anything but "the input ends inside a coded data set" is its call
    // (m == 0: the RSI the walk resumed in does not end inside the input itself)
    const uint64_t pos0 = m ? start_bit + sidx[m - 1u] : start_bit;
    BitReaderT<QuadFetch> br;
    br.init(QuadFetch{words, nwords}, s.end_bit, pos0);
    uint32_t b = m ? 0u : start_block, st = DEC_OK;
    uint64_t good = pos0;
    while (b < c.rsi) {
        uint32_t nblk = 1;
        st = skip_cds(br, c, (b == 0u && (c.flags & F_PREPROCESS)) ? 1u : 0u, b, nblk);
        if (st != DEC_OK) break;
        good = br.pos;
        b += nblk;
    }
    if (st != DEC_NEED_INPUT) return;
    res->n_rsi = base + m - 1u;
    res->tail_blocks = b;
    res->end_bit = good;
    res->status = DEC_OK;
    res->pad = 1u;
    res->bad_rsi = ~0ull;
    if (tail_slot) rsi_off[max_rsi] = m ? pos0 : before;
    __threadfence();
    flags[0] = 1u;
}

// (a walk that resumes inside an RSI: that RSI is number 0 and began at rsi_start, the first RSI start the chain meets is
// number 1 -- as k_index counts)
// (skip_if: the scheme in front has delivered -- no piece runs, and the serial walker behind the pieces returns at once)
__global__ void k_small_begin(uint32_t *flags, SmCursor *cur, uint64_t start_bit, uint64_t rsi_start, uint32_t start_block,
                              uint64_t *rsi_off, uint64_t max_rsi, const uint32_t *skip_if)
{
    const bool skip = skip_if && *skip_if;
    flags[0] = skip ? 1u : 0u;
    cur->bit = start_bit;
    cur->idx = start_block ? 1u : 0u;
    if (start_block && max_rsi && !skip) rsi_off[0] = rsi_start;
    cur->last_start = rsi_start;
    cur->stop = skip ? 1u : 0u;
    cur->pad = 0u;
}

static void launch_index_small(const Cfg &c, const SmallPlan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                               uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                               uint8_t *base, uint64_t rsi_start, uint32_t tail_slot, uint32_t start_block = 0,
                               const uint32_t *skip_if = nullptr)
{
    const TrStream s{words, nwords, end_bit};
    uint32_t *flags = reinterpret_cast<uint32_t *>(base + p.o_flags);
    SmCursor *cur = reinterpret_cast<SmCursor *>(base + p.o_flags + 64);
    uint16_t *e0 = reinterpret_cast<uint16_t *>(base + p.o_e0), *e1 = reinterpret_cast<uint16_t *>(base + p.o_e1);
    uint32_t *j[2] = {reinterpret_cast<uint32_t *>(base + p.o_ja), reinterpret_cast<uint32_t *>(base + p.o_jb)};
    uint32_t *sidx = reinterpret_cast<uint32_t *>(base + p.o_s);
    uint32_t *hop = reinterpret_cast<uint32_t *>(base + p.o_h), *hop2 = reinterpret_cast<uint32_t *>(base + p.o_h2);
    hipLaunchKernelGGL(k_small_begin, dim3(1), dim3(1), 0, st, flags, cur, start_bit, rsi_start, start_block, d_rsi_off, max_rsi,
                       skip_if);
    // (capped only where the scheme is enqueued as a fallback: a workgroup per 256 bits, retiring as it goes, is 10 - 18 %
    // faster than striding ones where the scheme does run -- 1 MiB streams: 0.37 against 0.47 ms)
    auto capped = [&](uint32_t g, uint32_t most) { return (!skip_if || g < most) ? g : most; };
    const uint32_t grid = capped((p.nbits + 1u + 255u) / 256u, kSmGrid);
    const uint32_t rgrid = (p.nbits + 1u + kSmRsiWg - 1u) / kSmRsiWg;
    const uint32_t sgrid = capped((p.scap + kSmRsiWg - 1u) / kSmRsiWg > rgrid ? (p.scap + kSmRsiWg - 1u) / kSmRsiWg : rgrid, kSmGrid / 4u);
    for (uint32_t piece = 0; piece < p.npieces; piece++) {
        hipLaunchKernelGGL(k_small_parse, dim3(grid), dim3(256), 0, st, c, s, (const SmCursor *)cur, p.nbits, e0, e1);
        if (p.hops)
            hipLaunchKernelGGL(k_small_hop, dim3(grid), dim3(256), 0, st, c, (const SmCursor *)cur, end_bit, p.nbits,
                               (const uint16_t *)e0, hop);
        if (p.hops > 1u)
            hipLaunchKernelGGL(k_small_hop2, dim3(grid), dim3(256), 0, st, (const SmCursor *)cur, end_bit, p.nbits,
                               (const uint32_t *)hop, hop2);
        hipLaunchKernelGGL(k_small_rsi, dim3(sgrid), dim3(kSmRsiWg), 0, st, c, (const SmCursor *)cur, end_bit, p.nbits,
                           (const uint16_t *)e0, (const uint16_t *)e1, (const uint32_t *)(p.hops ? hop : nullptr),
                           (const uint32_t *)(p.hops > 1u ? hop2 : nullptr), j[0], sidx, p.scap, start_block);
        for (uint32_t k = 0; k < p.levels; k++) {
            const uint32_t quarter = 1u << (2u * k), last = k + 1u == p.levels ? 1u : 0u;
            const uint32_t gq = capped((quarter + 255u) / 256u, kSmGrid);
            const uint32_t g = last ? gq : (gq > grid ? gq : grid);
            hipLaunchKernelGGL(k_small_double, dim3(g), dim3(256), 0, st, (const SmCursor *)cur, end_bit, p.nbits,
                               (const uint32_t *)j[k & 1u], j[(k & 1u) ^ 1u], sidx, quarter, p.scap, last);
        }
        if (tune("AEC_IDX_SMALL_FINISH", 1))
            hipLaunchKernelGGL(k_small_finish, dim3(1), dim3(1024), 0, st, c, s, cur, p.nbits, (const uint32_t *)sidx, p.scap, words,
                               nwords, d_rsi_off, max_rsi, d_res, tail_slot, flags, start_block);
    }
    // whatever was not delivered: the serial walker, which returns at once otherwise
    hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res,
                       (const uint64_t *)nullptr, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u, 1u, start_block, rsi_start, tail_slot,
                       TwTables{}, (ChunkEntry *)nullptr, SparseTables{}, (uint32_t *)nullptr, (uint64_t)0,
                       (const uint32_t *)flags);
}

}  // namespace

// (tuning build: the A/B switches that live in device globals)
static void idx_tuning_sync()
{
#ifdef AEC_TUNING
    static const int no_stretch = tune("AEC_IDX_STRETCH", 1) == 0 ? 1 : 0;
    static std::once_flag once[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [] { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_idx_no_stretch), &no_stretch, sizeof(int)); });
#endif
}

// The every-bit scheme as the FALLBACK behind a scheme that may give a stream up (round 6; tests/fuzz_index_gpu.py --time
// listed thirteen shapes of 300 below 0.6 GB/s, all of them such streams):
//  * behind the chains by plausibility (lock_plan mode 1), in place of the 64 agreeing chains and the trunk, which took
//    57 and 115 ms for 4 and 16 MiB of data in blocks of 64 with RSIs of 5 and 1 blocks whose guesses were judged wrong
//    (the every-bit pieces: 2.0 and 4.9 ms) -- streams of up to 2^28 bits, of up to 2^25 where RSIs have 64 blocks and more;
//  * behind the window tables of a stream of one piece (2^24 bits), whose walker gives up after kSparseSerialCap blocks
//    walked serially (k_index: serial_cap).
// 256 bytes in front of its workspace hold the flag "the scheme in front has delivered".
constexpr uint32_t kSparseSerialCap = 2048;
static SmallPlan small_fallback_plan(const Cfg &c, uint64_t total_bits, uint64_t max_rsi, uint32_t start_block,
                                     uint64_t rsi_bits_hint, bool behind_tables)
{
    if (!tune("AEC_IDX_SMALL_FALLBACK", 1)) return SmallPlan{};
    // (behind the chains by plausibility: RSIs of 64 blocks and more are the trunk's from 4 MiB of stream on -- its
    // block counts take 27 MiB with rsi 64 in 3 ms, the pieces 21 -- and below that its dozen launches cost more than
    // the pieces: 4 MiB of 16-bit data with rsi 256: 9.9 against 3.0 ms)
    const uint64_t most = behind_tables ? kSmMaxBits
                          : (c.rsi < 64u ? (uint64_t)tune("AEC_IDX_SMALL_FB_BITS", 1u << 28) : (uint64_t)tune("AEC_IDX_SMALL_FB_BITS_LONG", 1u << 25));
    if (total_bits > most) return SmallPlan{};
    return small_plan(c, total_bits, max_rsi, start_block, rsi_bits_hint, true);
}

int index_scheme(const Cfg &c, size_t in_bytes, uint64_t rsi_bits_hint, uint32_t start_block)
{
    const uint64_t bits = (uint64_t)in_bytes * 8;
    if (!bits) return 0;
    if (small_plan(c, bits, 1ull << 62, start_block, rsi_bits_hint).ok) return 4;
    if (region_plan(c, bits, rsi_bits_hint, false).ok) return 5;
    if (lock_plan(c, bits, rsi_bits_hint, start_block).ok) return 1;
    if (sparse2_plan(c, bits, rsi_bits_hint).ok) return 2;
    return trunk_plan(c, bits, rsi_bits_hint, 0).ok ? 3 : 0;
}


size_t index_workspace_bytes(const Cfg &c, size_t in_bytes, uint64_t start_bit, uint64_t rsi_bits_hint)
{
    const uint64_t end_bit = (uint64_t)in_bytes * 8;
    if (start_bit >= end_bit) return 0;
    // (a small stream: the brute-force scheme for a walk from an RSI start, the others for one that resumes inside)
    const SmallPlan sm = small_plan(c, end_bit - start_bit, 1ull << 62, 0u, rsi_bits_hint);
    // (the regions in front of whatever would run without them)
    const RegionPlan rp = region_plan(c, end_bit - start_bit, rsi_bits_hint, decode_bare_supported(c));
    const size_t rest = index_workspace_bytes_large(c, in_bytes, start_bit, rsi_bits_hint) + (rp.ok ? rp.bytes : 0);
    return sm.ok && sm.bytes > rest ? sm.bytes : rest;
}

size_t index_workspace_bytes_large(const Cfg &c, size_t in_bytes, uint64_t start_bit, uint64_t rsi_bits_hint)
{
    const uint64_t end_bit = (uint64_t)in_bytes * 8;
    if (start_bit >= end_bit) return 0;
    const LockPlan lp = lock_plan(c, end_bit - start_bit, rsi_bits_hint, 0u);
    if (lp.ok && lp.mode == 0u) return lp.bytes;
    if (lp.ok) {                                           // (mode 1: + what runs behind it for the streams it abandons)
        const SmallPlan fb = small_fallback_plan(c, end_bit - start_bit, 1ull << 62, 0u, rsi_bits_hint, false);
        if (fb.ok) return lp.bytes + 256 + fb.bytes;
        const LockPlan l0 = lock_plan_alt(c, end_bit - start_bit, rsi_bits_hint, 0u);
        const TrunkPlan tp = trunk_plan(c, end_bit - start_bit, rsi_bits_hint, 0);
        return lp.bytes + (l0.ok ? l0.bytes : 0) + (tp.ok ? tp.bytes : 0);
    }
    const Sparse2Plan sp = sparse2_plan(c, end_bit - start_bit, rsi_bits_hint);
    if (sp.ok) {
        // (many spans of windows: two sets of tables, so that the spans can be pipelined -- launch_index_sparse)
        const uint64_t span = (uint64_t)sp.nwin_max * sp.g.core;
        const SmallPlan fb = small_fallback_plan(c, end_bit - start_bit, 1ull << 62, 0u, rsi_bits_hint, true);
        return (end_bit - start_bit > (kS2PipeSpans - 1) * span ? 2 * sp.bytes : sp.bytes) + (fb.ok ? 256 + fb.bytes : 0);
    }
    const TrunkPlan p = trunk_plan(c, end_bit - start_bit, rsi_bits_hint, 0);
    return p.ok ? p.bytes : 0;
}

bool launch_index(const Cfg &c, const uint8_t *d_in, size_t in_bytes, uint64_t start_bit,
                  uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                  void *d_ws, size_t ws_bytes, uint64_t rsi_bits_hint, uint32_t start_block, uint64_t rsi_start,
                  uint32_t tail_slot, uint64_t *d_seg_bits, uint64_t stop_near)
{
    const uint32_t *words = reinterpret_cast<const uint32_t *>(d_in);
    const uint64_t nwords = (in_bytes + 3) / 4, end_bit = (uint64_t)in_bytes * 8;
    idx_tuning_sync();
    // Low-entropy streams whose RSIs fit a window: candidates and RSI hypotheses per window (k_spec2); everything
    // else: the trunk.
    // a small stream (a chunk of a dataset): every bit parsed, the RSI starts by pointer doubling
    if (d_ws && ws_bytes && start_bit < end_bit && !d_seg_bits && !stop_near) {
        const SmallPlan sm = small_plan(c, end_bit - start_bit, max_rsi, start_block, rsi_bits_hint);
        if (sm.ok && ws_bytes >= sm.bytes) {
            launch_index_small(c, sm, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                               static_cast<uint8_t *>(d_ws), rsi_start, tail_slot, start_block);
            return false;
        }
    }
    // a large stream: regions walked from guessed entries; the schemes below are enqueued behind and return at once
    // where it has delivered
    const uint32_t *done = nullptr;
    bool segs_filled = false;
    if (d_ws && ws_bytes && start_bit < end_bit) {
        const RegionPlan rp = region_plan(c, end_bit - start_bit, rsi_bits_hint, d_seg_bits != nullptr);
        if (rp.ok && ws_bytes >= rp.bytes) {
            done = launch_index_regions(c, rp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                                        static_cast<uint8_t *>(d_ws), start_block, rsi_start, tail_slot, d_seg_bits);
            d_ws = static_cast<uint8_t *>(d_ws) + rp.bytes;
            ws_bytes -= rp.bytes;
            segs_filled = d_seg_bits != nullptr;
        }
    }
    if (d_ws && ws_bytes && start_bit < end_bit && !d_seg_bits) {
        const LockPlan lp = lock_plan(c, end_bit - start_bit, rsi_bits_hint, start_block);
        if (lp.ok && lp.mode == 0u && ws_bytes >= lp.bytes) {
            launch_index_locked(c, lp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                                static_cast<uint8_t *>(d_ws), start_block, rsi_start, tail_slot, true, done);
            return false;
        }
        if (lp.ok && lp.mode == 1u && ws_bytes >= lp.bytes) {
            // entries by plausibility, the exact machinery of the phase-locked scheme behind them -- and behind that, for
            // the streams whose options say nothing, what would have run without: the 64 agreeing chains where RSIs are
            // short, then the trunk; every kernel of a later scheme returns at once if the stream has been delivered
            uint8_t *wb = static_cast<uint8_t *>(d_ws);
            size_t used = lp.bytes;
            // (streams of up to 2^28 bits: the every-bit scheme piece by piece in place of both -- small_fallback_plan)
            const SmallPlan fb = stop_near ? SmallPlan{}
                                           : small_fallback_plan(c, end_bit - start_bit, max_rsi, start_block, rsi_bits_hint, false);
            if (fb.ok && ws_bytes >= used + 256 + fb.bytes) {
                launch_index_locked(c, lp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb, start_block,
                                    rsi_start, tail_slot, false, done);
                launch_index_small(c, fb, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb + used + 256,
                                   rsi_start, tail_slot, start_block, reinterpret_cast<const uint32_t *>(wb + lp.o_flags));
                return false;
            }
            const LockPlan l0 = lock_plan_alt(c, end_bit - start_bit, rsi_bits_hint, start_block);
            const bool have0 = l0.ok && ws_bytes >= used + l0.bytes;
            const size_t off0 = used;
            if (have0) used += l0.bytes;
            const TrunkPlan tp = trunk_plan(c, end_bit - start_bit, rsi_bits_hint, ws_bytes - used);
            launch_index_locked(c, lp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb, start_block,
                                rsi_start, tail_slot, !tp.ok && !have0, done);
            done = reinterpret_cast<const uint32_t *>(wb + lp.o_flags);
            if (have0) {
                launch_index_locked(c, l0, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb + off0,
                                    start_block, rsi_start, tail_slot, !tp.ok, done);
                done = reinterpret_cast<const uint32_t *>(wb + off0 + l0.o_flags);
            }
            if (tp.ok)
                launch_index_trunk(c, tp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb + used,
                                   start_block, rsi_start, tail_slot, nullptr, done);
            return false;
        }
    }
    if (d_ws && ws_bytes && start_bit < end_bit) {
        const Sparse2Plan sp = sparse2_plan(c, end_bit - start_bit, rsi_bits_hint);
        const SmallPlan fb = (sp.ok && !stop_near && !done)
                                 ? small_fallback_plan(c, end_bit - start_bit, max_rsi, start_block, rsi_bits_hint, true)
                                 : SmallPlan{};
        if (sp.ok && fb.ok && ws_bytes >= sp.bytes + 256 + fb.bytes &&
            end_bit - start_bit / sp.g.core * sp.g.core <= (uint64_t)sp.nwin_max * sp.g.core) {
            // (one span of tables: the walker may give the stream up, the every-bit scheme behind it takes it then)
            uint8_t *wb = static_cast<uint8_t *>(d_ws);
            uint32_t *delivered = reinterpret_cast<uint32_t *>(wb + sp.bytes);
            (void)hipMemsetAsync(delivered, 0, 4, st);
            launch_index_sparse(c, sp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb, sp.bytes,
                                start_block, rsi_start, tail_slot, 0ull, nullptr, (uint32_t)tune("AEC_IDX_SPARSE_SERIAL_CAP", kSparseSerialCap),
                                delivered);
            launch_index_small(c, fb, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st, wb + sp.bytes + 256,
                               rsi_start, tail_slot, start_block, delivered);
            return segs_filled;
        }
        if (sp.ok && ws_bytes >= sp.bytes) {
            launch_index_sparse(c, sp, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                                static_cast<uint8_t *>(d_ws), ws_bytes, start_block, rsi_start, tail_slot, stop_near, done);
            // (segment starts the regions did not deliver stay ~0: such RSIs are decoded by one lane each)
            return segs_filled;
        }
    }
    TrunkPlan p{};
    if (d_ws && ws_bytes && start_bit < end_bit) p = trunk_plan(c, end_bit - start_bit, rsi_bits_hint, ws_bytes);
    if (!p.ok) {                                       // serial walk only
        hipLaunchKernelGGL(k_index, dim3(1), dim3(64), 0, st, c, words, nwords, end_bit, start_bit, d_rsi_off,
                           max_rsi, d_res, (const uint64_t *)nullptr, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u,
                           1u, start_block, rsi_start, tail_slot, TwTables{}, (ChunkEntry *)nullptr, SparseTables{},
                           (uint32_t *)nullptr, (uint64_t)0, done);
        return segs_filled;
    }
    // (segment starts: the caller has set the table to ~0; the RSI the walk resumes in has none -- its first
    // blocks lie in front of the walk)
    launch_index_trunk(c, p, words, nwords, end_bit, start_bit, d_rsi_off, max_rsi, d_res, st,
                       static_cast<uint8_t *>(d_ws), start_block, rsi_start, tail_slot, d_seg_bits, done);
    return d_seg_bits != nullptr;
}

// Batch over the window tables: how many hops a stream of max_chunk_bytes may take, and the workspace
static uint32_t batch_hop_cap(const Sparse2Plan &p, size_t max_chunk_bytes)
{
    return (uint32_t)(2 * ((uint64_t)max_chunk_bytes * 8 / p.g.core + 2) + 8);
}

size_t index_batch_workspace_bytes(const Cfg &c, size_t in_bytes, uint64_t n_chunks, size_t max_chunk_bytes,
                                   uint64_t rsi_bits_hint)
{
    // (short RSIs: the window tables resolve none of them -- chunk by chunk through launch_index, whose phase-locked
    // chains do)
    if (c.rsi <= kLockMaxRsi && (c.flags & F_PREPROCESS) && !(c.flags & F_PAD_RSI)) return 0;
    const Sparse2Plan p = sparse2_plan(c, (uint64_t)in_bytes * 8, rsi_bits_hint);
    if (!p.ok || n_chunks == 0) return 0;
    const uint64_t nwin = ((uint64_t)in_bytes * 8 + p.g.core - 1) / p.g.core;
    if (nwin > p.nwin_max) return 0;                       // (more than one span of tables: the caller splits the batch)
    return p.bytes + (((size_t)n_chunks * batch_hop_cap(p, max_chunk_bytes) * sizeof(IdxHop) + 255) & ~(size_t)255) +
           (((size_t)n_chunks * 4 + 255) & ~(size_t)255);
}

// d_ws (optional, index_batch_workspace_bytes()): the streams are low-entropy and long enough for the window
// tables -- ONE speculation launch over the whole buffer (every stream's start a forced candidate), then one
// wavefront per stream hops over the tables instead of walking coded data set by coded data set.
void launch_index_batch(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_chunk_off,
                        uint64_t n_chunks, uint64_t rsi_per_chunk, uint64_t *d_rsi_off, DecResult *d_res,
                        hipStream_t st, void *d_ws, size_t ws_bytes, size_t max_chunk_bytes, uint64_t rsi_bits_hint)
{
    if (n_chunks == 0) return;
    idx_tuning_sync();
    const uint32_t *words = reinterpret_cast<const uint32_t *>(d_in);
    const uint64_t nwords = (in_bytes + 3) / 4, end_bit = (uint64_t)in_bytes * 8;
    const size_t need = d_ws ? index_batch_workspace_bytes(c, in_bytes, n_chunks, max_chunk_bytes, rsi_bits_hint) : 0;
    if (!need || ws_bytes < need) {
        hipLaunchKernelGGL(k_index, dim3((uint32_t)n_chunks), dim3(64), 0, st, c, words, nwords, end_bit, (uint64_t)0,
                           d_rsi_off, rsi_per_chunk, d_res, d_chunk_off, (IdxHop *)nullptr, 0u, (IdxCarry *)nullptr, 1u,
                           1u, 0u, (uint64_t)0, 0u, TwTables{}, (ChunkEntry *)nullptr, SparseTables{});
        return;
    }
    allow_big_lds2();
    const Sparse2Plan p = sparse2_plan(c, end_bit, rsi_bits_hint);
    uint8_t *base = static_cast<uint8_t *>(d_ws);
    const uint32_t nwin = (uint32_t)((end_bit + p.g.core - 1) / p.g.core);
    SparseTables t;
    t.bitmap = reinterpret_cast<const uint32_t *>(base + p.o_bitmap);
    t.pre = reinterpret_cast<const uint16_t *>(base + p.o_pre);
    t.rec = reinterpret_cast<const uint2 *>(base + p.o_rec);
    t.cpos = reinterpret_cast<const uint16_t *>(base + p.o_cpos);
    t.ccnt = reinterpret_cast<const uint32_t *>(base + p.o_ccnt);
    t.lo = 0;
    t.hi = (uint64_t)nwin * p.g.core;
    t.core = p.g.core;
    t.cap = p.g.cap_core;
    t.wide = nullptr;
    t.wpc = p.wpc;
    const uint32_t hop_cap = batch_hop_cap(p, max_chunk_bytes);
    IdxHop *hops = reinterpret_cast<IdxHop *>(base + p.bytes);
    uint32_t *nhops = reinterpret_cast<uint32_t *>(base + p.bytes +
                                                   (((size_t)n_chunks * hop_cap * sizeof(IdxHop) + 255) & ~(size_t)255));
    if (p.g.v4)
        hipLaunchKernelGGL(k_spec4, dim3(nwin), dim3(1024), p.lds, st, c, words, nwords, end_bit, (uint64_t)0, (uint64_t)0, p.g,
                           const_cast<uint32_t *>(t.bitmap), const_cast<uint16_t *>(t.pre), const_cast<uint2 *>(t.rec),
                           const_cast<uint16_t *>(t.cpos), const_cast<uint32_t *>(t.ccnt), (unsigned long long *)nullptr,
                           d_chunk_off, (uint32_t)n_chunks);
    else
        hipLaunchKernelGGL(k_spec2, dim3(nwin), dim3(1024), p.lds, st, c, words, nwords, end_bit, (uint64_t)0, (uint64_t)0, p.g,
                           const_cast<uint32_t *>(t.bitmap), const_cast<uint16_t *>(t.pre), const_cast<uint2 *>(t.rec),
                           const_cast<uint16_t *>(t.cpos), const_cast<uint32_t *>(t.ccnt), (unsigned long long *)nullptr,
                           d_chunk_off, (uint32_t)n_chunks);
    hipLaunchKernelGGL(k_index, dim3((uint32_t)n_chunks), dim3(64), 0, st, c, words, nwords, end_bit, (uint64_t)0, d_rsi_off,
                       rsi_per_chunk, d_res, d_chunk_off, hops, hop_cap, (IdxCarry *)nullptr, 1u, 1u, 0u, (uint64_t)0, 0u,
                       TwTables{}, (ChunkEntry *)nullptr, t, nhops);
    hipLaunchKernelGGL(k_expand2, dim3((uint32_t)(((uint64_t)n_chunks * hop_cap + 255) / 256)), dim3(256), 0, st, t,
                       (const IdxCarry *)nullptr, hops, nhops, (uint32_t)n_chunks, hop_cap, d_rsi_off, rsi_per_chunk);
}

}  // namespace aec
