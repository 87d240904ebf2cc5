// aec_region.hip -- the REGION index: RSI starts of a large bare stream by lanes that walk regions from guessed entries
// (per-lane arithmetic, the guess and the argument why nothing delivered rests on it: aec_region.h).
//
// A bare stream has no entry points and the reference walks it coded data set by coded data set (src/decode.c:402-421).
// The older schemes of aec_idx.hip try RSI starts as HYPOTHESES at every boundary a chain visits and tabulate the
// outcome -- exact without a guess, but hundreds of parses per coded data set of the stream (config 2: 73 ms per 4 GiB,
// 0.0013 of the HBM roofline).  A lane that KNOWS where an RSI starts walks on from there at one parse per coded data
// set.  So:
//   k_rg_guess    a lane per region: the first RSI start behind the region's first bit, recognised by the options of
//                 the coded data sets around it (RgGuess);
//   k_rg_link     which guesses are kept (entries in increasing order), how many: too few and the scheme steps aside;
//   k_rg_walk     a lane per region: the walk with the RSI's bookkeeping from the entry to the first RSI start at or
//                 behind the next region's entry; counts the RSI starts on the way;
//   k_rg_mend     x passes: a region whose entry is not where the walk in front of it arrived (and whose predecessor
//                 is not in doubt itself) takes that place as its entry and is walked again;
//   k_rg_scan     do all entries agree with the walks in front of them, up to the region where the input ended?  RSI
//                 starts in front of every region;
//   k_rg_fill     the walks once more: RSI starts (and segment starts) into the caller's tables, the result record.
// Region 0 starts on the caller's exact state; an entry that equals the exit of an exact walk is exact; k_rg_scan
// delivers only if that holds for every region.  Otherwise nothing is written to the record, flags[0] stays 0 and
// the schemes of aec_idx.hip, enqueued behind with this flag as their skip_if, take the stream as before.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <vector>

#include "aec_kernels.h"
#include "aec_region.h"
#include "aec_tune.h"

namespace aec {

namespace {

struct RgTables {
    RgEntry *found;            // [nreg] the guesses
    RgEntry *entry[2];         // [nreg] entries, double buffered over the mending passes
    RgState *exit[2];          // [nreg] where the walk of the region arrived
    uint32_t *cnt[2];          // [nreg] RSI starts the walk met
    uint64_t *base;            // [nreg + 1] ... in front of the region
    uint64_t *list;            // [nreg * K] the first K of them, as the region's last walk met them
    uint64_t *slist;           // [nreg * K * segments per RSI] and their segment starts (null: not asked for)
    uint32_t K;
    uint32_t *flags;           // [0] delivered, [1] entries do not agree, [2] region in which the input ended,
                               // [3] stepped aside, [4] live regions, [5] regions mended (all passes), [6] the guesses' queue,
                               // [7] first region whose entry is not where the walk in front arrived
    uint32_t nreg, budget, period;
    uint64_t region_bits, lo, max_walk;
    const uint32_t *skip_if;
};

// the lane's ring: a column of the workgroup's LDS (one wavefront per workgroup)
#define RG_RING(ps, W)                                                          \
    extern __shared__ __attribute__((aligned(16))) uint32_t rg_lds[];          \
    RgRingT<64u, W> ps{s, c};                                                   \
    ps.init(rg_lds + (threadIdx.x & 63u), t.period)
constexpr size_t rg_lds_bytes(uint32_t words) { return (size_t)rg_ring_rows(words) * 64u * 4u; }

__device__ __forceinline__ bool rg_off(const RgTables &t)
{
    return (t.skip_if && *t.skip_if) || t.flags[3];
}

// A lane takes regions from a queue (flags[6]) until it is empty: the guesses are of very different lengths -- a third of
// the budget on average, the whole of it for a few -- and a wavefront that held 64 of them would wait for its slowest.
__global__ void __launch_bounds__(64)
k_rg_guess(const Cfg c, const TrStream s, const RgTables t)
{
    if (t.skip_if && *t.skip_if) return;
    RG_RING(ps, 64u);
    const uint32_t lane = threadIdx.x & 63u;
    RgGuess g;
    g.init(0);
    uint32_t r = 0, r0 = 0;
    const uint32_t limit = (uint32_t)(2u * t.region_bits);
    bool have = false, dry = false;
    for (;;) {
        const uint64_t want = __ballot(!have && !dry);
        if (want) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(want), n = (uint32_t)__popcll(want);
            uint32_t first = 0;
            if (lane == leader) first = atomicAdd(&t.flags[6], n);
            first = (uint32_t)__shfl((int)first, (int)leader);
            if (!have && !dry) {
                r = 1u + first + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                if (r < t.nreg) {
                    const uint64_t from = t.lo + (uint64_t)r * t.region_bits;
                    ps.seat(from);
                    r0 = ps.rel_of(from);
                    g.init(r0);
                    if (!(r0 < ps.end_rel && ps.end_rel - r0 > c.id_len)) g.mode = RgGuess::NONE;
                    have = true;
                } else {
                    dry = true;
                }
            }
        }
        if (!__any(have)) break;
        if (have) {
            if (g.busy() && g.parses < t.budget && !(g.mode <= RgGuess::WALK && g.q - r0 >= limit)) {
                uint32_t id, nz;
                const uint32_t len = ps.cds(g.q, g.ref, id, nz);
                g.step(c, ps.end_rel, len, id, nz);
            } else {
                const bool got = g.mode == RgGuess::FOUND;
                t.found[r] = RgEntry{got ? ps.pos_of(g.found) : 0u, 0u, got ? 1u : 0u};
                have = false;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_rg_link(const RgTables t, uint64_t start_bit, uint32_t start_block)
{
    if (t.skip_if && *t.skip_if) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t live = 0;
    if (r < t.nreg) {
        RgEntry e;
        if (r == 0u) {
            e = RgEntry{start_bit, start_block, 1u};
        } else {
            e = t.found[r];
            const bool have_next = r + 1u < t.nreg;
            e.live = rg_keep(e, have_next ? t.found[r + 1u] : RgEntry{0u, 0u, 0u}, have_next) ? 1u : 0u;
        }
        t.entry[0][r] = e;
        live = e.live;
    }
    const uint32_t n = (uint32_t)__popcll(__ballot(live != 0u));
    if ((threadIdx.x & 63u) == 0u && n) atomicAdd(&t.flags[4], n);
}

// too few entries: the options of this data say nothing (or the stream is a run of constants, a few coded data sets
// per region) -- the walks would be long and few; the schemes behind take the stream
__global__ void k_rg_judge(const RgTables t)
{
    if (t.skip_if && *t.skip_if) return;
    if (t.flags[4] * 2u < t.nreg) t.flags[3] = 1u;
}

__device__ __forceinline__ uint32_t rg_next_live(const RgEntry *e, uint32_t r, uint32_t nreg)
{
    uint32_t q = r + 1u;
    while (q < nreg && !e[q].live) q++;
    return q;
}
__device__ __forceinline__ uint32_t rg_prev_live(const RgEntry *e, uint32_t r)
{
    uint32_t q = r;
    while (q-- > 0u)
        if (e[q].live) return q;
    return 0u;
}
__device__ __forceinline__ bool rg_differs(const RgEntry *e, const RgState *ex, uint32_t r)
{
    const RgState p = ex[rg_prev_live(e, r)];
    const RgEntry m = e[r];
    return p.st != 0u || p.pos != m.pos || p.b != m.b;
}

template <uint32_t W>
__device__ __forceinline__ void rg_walk_region(const Cfg &c, const TrStream &s, const RgTables &t, const RgEntry *e, uint32_t r,
                                               RgEntry mine, RgState *ex_out, uint32_t *cnt_out)
{
    RG_RING(ps, W);
    RgState x{mine.pos, mine.b, 0u};
    const uint32_t nl = rg_next_live(e, r, t.nreg);
    uint32_t n = 0;
    const uint32_t spr = c.segs_per_rsi;
    uint64_t *list = t.list + (size_t)r * t.K, *slist = t.slist ? t.slist + (size_t)r * t.K * spr : nullptr;
    // (the RSI starts go to the region's list as they are met, so that the last pass need not walk again: k_rg_fill)
    rg_walk(ps, c, x, nl < t.nreg ? e[nl].pos : ~0ull, nl < t.nreg ? t.max_walk : ~0ull,
            [&](uint64_t pos) {
                if (n < t.K) {
                    list[n] = pos;
                    if (slist) {
                        slist[(size_t)n * spr] = pos;
                        for (uint32_t k = 1; k < spr; k++) slist[(size_t)n * spr + k] = ~0ull;
                    }
                }
                n++;
                return true;
            },
            [&](uint32_t b, uint64_t pos) {
                // (the RSI a walk resumed in has no entries: its first blocks lie in front of the input)
                if (slist && n && n <= t.K) slist[(size_t)(n - 1u) * spr + (b >> 6)] = pos;
            });
    cnt_out[r] = n;
    ex_out[r] = x;
}

// W: the lanes' rings (32 words where coded data sets are short: twice the wavefronts per CU)
template <uint32_t W>
__global__ void __launch_bounds__(64)
k_rg_walk(const Cfg c, const TrStream s, const RgTables t)
{
    if (rg_off(t)) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nreg) return;
    const RgEntry mine = t.entry[0][r];
    if (!mine.live) {
        t.exit[0][r] = RgState{0u, 0u, 0u};
        t.cnt[0][r] = 0u;
        return;
    }
    rg_walk_region<W>(c, s, t, t.entry[0], r, mine, t.exit[0], t.cnt[0]);
}

// one mending pass, from the tables `cur` into the tables `cur ^ 1`
template <uint32_t W>
__global__ void __launch_bounds__(64)
k_rg_mend(const Cfg c, const TrStream s, const RgTables t, uint32_t cur)
{
    if (rg_off(t)) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nreg) return;
    const RgEntry *e = t.entry[cur];
    const RgState *ex = t.exit[cur];
    RgEntry mine = e[r];
    bool mend = false;
    if (r && mine.live && rg_differs(e, ex, r)) {
        const uint32_t pr = rg_prev_live(e, r);
        // (not from a walk that ended, and not from a region that is in doubt itself: the exit of a walk from a wrong
        // entry has the wrong count of blocks for good, and walking on from it throws a right guess after a wrong one)
        mend = ex[pr].st == 0u && !(pr != 0u && rg_differs(e, ex, pr));
        if (mend) mine = RgEntry{ex[pr].pos, ex[pr].b, 1u};
    }
    t.entry[cur ^ 1u][r] = mine;
    if (!mend) {
        t.exit[cur ^ 1u][r] = ex[r];
        t.cnt[cur ^ 1u][r] = t.cnt[cur][r];
        return;
    }
    atomicAdd(&t.flags[5], 1u);
    // (the entries of the regions behind are the same in both sets unless they are mended in this pass -- and then the one
    // in front of them was in doubt: the walk's target is the entry the NEXT pass will compare with)
    rg_walk_region<W>(c, s, t, e, r, mine, t.exit[cur ^ 1u], t.cnt[cur ^ 1u]);
}

__global__ void k_rg_prep(const RgTables t)
{
    t.flags[2] = t.nreg;
    t.flags[7] = t.nreg;
}

// the first region in which the walk ended (flags[2]), the first whose entry is not where the walk in front arrived (flags[7])
__global__ void __launch_bounds__(256)
k_rg_check(const RgTables t, uint32_t cur)
{
    if (rg_off(t)) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t.nreg) return;
    const RgEntry *e = t.entry[cur];
    const RgState *ex = t.exit[cur];
    if (!e[r].live) return;
    if (ex[r].st) atomicMin(&t.flags[2], r);
    if (r && rg_differs(e, ex, r)) atomicMin(&t.flags[7], r);
}

// RSI starts per wavefront of regions (the regions behind the one where the input ended count nothing)
__global__ void __launch_bounds__(64)
k_rg_sums(const RgTables t, uint32_t cur)
{
    if (rg_off(t)) return;
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    const uint32_t fe = t.flags[2];
    uint32_t v = 0;
    if (r < t.nreg) {
        const bool live = t.entry[cur][r].live && r <= fe;
        if (!live) t.cnt[cur][r] = 0u;
        v = live ? t.cnt[cur][r] : 0u;
    }
#pragma unroll
    for (uint32_t d = 32; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, (int)d);
    if (threadIdx.x == 0) t.base[blockIdx.x] = v;
}

// one workgroup: RSI starts in front of every wavefront of regions (in place), and the verdict
__global__ void __launch_bounds__(1024)
k_rg_scan(const RgTables t, uint32_t cur)
{
    if (rg_off(t)) return;
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nw = (t.nreg + 63u) / 64u;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < nw; i0 += 1024u) {
        const uint32_t i = i0 + tid;
        const uint64_t v = i < nw ? t.base[i] : 0u;
        uint64_t inc = v;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint64_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        if (i < nw) t.base[i] = before + inc - v;
        __syncthreads();
        if (tid == 1023u) carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) {
        // (a walk must have ended -- the last region's has no target -- and only "the input ended" is this scheme's to
        // report; every entry up to there must be where the walk in front of it arrived)
        const uint32_t fe = t.flags[2];
        t.flags[1] = (fe >= t.nreg || t.flags[7] <= fe || t.exit[cur][fe].st != 1u) ? 1u : 0u;
    }
}

// the RSI starts (and segment starts), and the result record: written by the region in which the walk ended or the
// caller's bound was met (as k_lock_fill of aec_idx.hip, whose record conventions these are)
__global__ void __launch_bounds__(64)
k_rg_fill(const Cfg c, const TrStream s, const RgTables t, uint32_t cur, uint64_t *__restrict__ rsi_off, uint64_t max_rsi,
          DecResult *res, uint32_t tail_slot, uint64_t rsi_start_in, uint32_t start_block, uint64_t *__restrict__ seg_bits)
{
    if (rg_off(t)) return;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    // RSI starts in front of the region: in front of its wavefront of regions (k_rg_scan) + those of the lanes below
    uint32_t mycnt = r < t.nreg ? t.cnt[cur][r] : 0u, incl = mycnt;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if ((threadIdx.x & 63u) >= d) incl += o;
    }
    if (r >= t.nreg || t.flags[1] || r > t.flags[2]) return;
    const RgEntry *e = t.entry[cur];
    const RgEntry mine = e[r];
    if (!mine.live) return;
    const uint64_t base_r = t.base[r >> 6] + incl - mycnt;
    // A walk that resumes INSIDE an RSI (streaming callers: start_block blocks of it lie in front of the input): that
    // RSI is number 0 and began at rsi_start_in, the first RSI start the walk meets is number 1 (as k_index counts).
    const uint64_t off = start_block ? 1u : 0u;
    uint64_t idx = base_r + off;
    if (off && r == 0u && max_rsi) rsi_off[0] = rsi_start_in;
    if (idx > max_rsi) return;                       // (behind the caller's bound)
    const bool last = r == t.flags[2];
    const uint32_t nl = rg_next_live(e, r, t.nreg);
    const uint32_t spr = c.segs_per_rsi;
    if (mycnt <= t.K && (!seg_bits || t.slist)) {
        // the region's RSI starts lie in its list: no walk
        const uint64_t *list = t.list + (size_t)r * t.K;
        const uint64_t *slist = t.slist ? t.slist + (size_t)r * t.K * spr : nullptr;
        // the last RSI start in front of this region, from the lists of the regions in front (a region whose list is
        // not whole: the caller's, as a walk would find it -- only asked by the lane that writes the record)
        auto start_in_front = [&]() -> uint64_t {
            for (uint32_t q = r; q-- > 0u;) {
                if (!e[q].live || !t.cnt[cur][q]) continue;
                const uint32_t nq = t.cnt[cur][q];
                if (nq <= t.K) return t.list[(size_t)q * t.K + nq - 1u];
                break;
            }
            return rsi_start_in;
        };
        uint32_t i = 0;
        bool clipped = false;
        for (; i < mycnt; i++) {
            if (idx == max_rsi) {                    // the caller's bound: the pass ends on this RSI start
                clipped = true;
                break;
            }
            rsi_off[idx] = list[i];
            if (seg_bits)
                for (uint32_t k = 0; k < spr; k++) seg_bits[idx * spr + k] = slist[(size_t)i * spr + k];
            idx++;
        }
        const RgState x = t.exit[cur][r];
        if (!clipped && !(last && x.st)) return;     // the walk goes on in the next region
        // (an RSI start the next region begins on is that region's: a bound met there is met there)
        if (clipped) {
            res->n_rsi = max_rsi;
            res->tail_blocks = 0;
            res->end_bit = list[i];
            res->status = DEC_OK;
            res->pad = 0u;
            res->bad_rsi = ~0ull;
            if (tail_slot) rsi_off[max_rsi] = i ? list[i - 1u] : start_in_front();
            __threadfence();
            t.flags[0] = 1u;
            return;
        }
        if (x.st != 1u || s.end_bit - x.pos > kTrMaxScan) return;
        {
            BitReader br;
            br.init(s.words, s.nwords, s.end_bit, x.pos);
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
        if (tail_slot) rsi_off[max_rsi] = mycnt ? list[mycnt - 1u] : start_in_front();
        __threadfence();
        t.flags[0] = 1u;
        return;
    }
    // a region with more RSI starts than its list holds (a constant stretch: an RSI in a few dozen bits): walked again
    RG_RING(ps, 64u);
    RgState x{mine.pos, mine.b, 0u};
    uint64_t cur_start = 0;
    bool met = false, clipped = false;
    rg_walk(ps, c, x, (nl < t.nreg && !last) ? e[nl].pos : ~0ull, ~0ull,
            [&](uint64_t pos) {
                if (idx == max_rsi) {                // the caller's bound: the pass ends on this RSI start
                    clipped = true;
                    return false;
                }
                rsi_off[idx] = pos;
                if (seg_bits) seg_bits[idx * spr] = pos;
                cur_start = pos;
                met = true;
                idx++;
                return true;
            },
            [&](uint32_t b, uint64_t pos) {
                // (the RSI the walk resumed in has no table entries: its first blocks lie in front of the input)
                if (seg_bits && idx > off) seg_bits[(idx - 1u) * spr + (b >> 6)] = pos;
            });
    if (!clipped && !x.st) return;                   // the walk goes on in the next region
    // the last RSI start in front of this region (the RSI the walk is in when it enters it): only the lane that ends
    // the walk asks, and only if it met none itself
    auto start_in_front = [&]() -> uint64_t {
        for (uint32_t q = r; q-- > 0u;) {
            if (!e[q].live || !t.cnt[cur][q]) continue;
            auto &pq = ps;                               // (this lane's own walk is over)
            RgState y{e[q].pos, e[q].b, 0u};
            uint64_t lastpos = rsi_start_in;
            const uint32_t qn = rg_next_live(e, q, t.nreg);
            rg_walk(pq, c, y, qn < t.nreg ? e[qn].pos : ~0ull, ~0ull,
                    [&](uint64_t pos) {
                        lastpos = pos;
                        return true;
                    },
                    [](uint32_t, uint64_t) {});
            return lastpos;
        }
        return rsi_start_in;
    };
    if (clipped) {
        res->n_rsi = max_rsi;
        res->tail_blocks = 0;
        res->end_bit = x.pos;
        res->status = DEC_OK;
        res->pad = 0u;
        res->bad_rsi = ~0ull;
        if (tail_slot) rsi_off[max_rsi] = met ? cur_start : start_in_front();
        __threadfence();
        t.flags[0] = 1u;
        return;
    }
    // the walk ended here: only "the input ends inside this coded data set", confirmed by the sequential reader, is
    // delivered; anything else is the serial walker's to report
    if (x.st != 1u || s.end_bit - x.pos > kTrMaxScan) return;
    {
        BitReader br;
        br.init(s.words, s.nwords, s.end_bit, x.pos);
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
    if (tail_slot) rsi_off[max_rsi] = met ? cur_start : start_in_front();
    __threadfence();
    t.flags[0] = 1u;
}

}  // namespace

RegionPlan region_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint, bool want_segments)
{
    RegionPlan p{};
    if (!tune("AEC_IDX_REGIONS", 1)) return p;
    // the guess looks for the coded data set that holds a reference sample and reads the options around it
    if (!(c.flags & F_PREPROCESS) || (c.flags & F_PAD_RSI) || c.id_len < 3u || !rsi_bits_hint) return p;
    if (index_incompressible(c, rsi_bits_hint)) return p;
    // (large streams: the passes are as long as one lane's guess and walks -- a few milliseconds whatever the size -- and
    // below half a gigabit of stream the window tables, which cost 3 ms per 100 MiB of input, are through first)
    if (total_bits < (uint64_t)tune("AEC_IDX_REGIONS_MIN", 1u << 29)) return p;
    // RSIs beyond the phase-locked scheme's
    const uint64_t cds = rsi_bits_hint / c.rsi;
    if (c.rsi < 48u || cds < 8u) return p;
    // regions: enough of them to fill the chip's lanes a few times over (the passes are as long as one lane's walk), but
    // of a few RSIs, 16 kbit at least: the guess walks an RSI or two whatever the region's size
    uint64_t region = (uint64_t)tune("AEC_IDX_REGION_BITS", 0);
    if (!region) {
        region = total_bits / (uint64_t)tune("AEC_IDX_REGION_LANES", 196608);
        const uint64_t rmin = rsi_bits_hint > 16384 ? rsi_bits_hint : 16384;
        if (region < rmin) region = rmin;
    }
    region = (region + 63) & ~63ull;
    const uint64_t nreg = (total_bits + region - 1) / region;
    // (long coded data sets -- config 3, the sample file: a region is an RSI and a lane's walk of it takes milliseconds
    // whatever the size of the stream; from a few thousand RSIs on that is less than the trunk's 20 ms per GiB)
    // (and only where RSIs have eight segments and more: shorter ones with long coded data sets -- the sample file -- are
    // the plausibility scheme's of aec_idx.hip, a wavefront per region: 16 against 24 ms per GiB)
    if (cds > 128u && c.segs_per_rsi < 8u) return p;
    if (nreg < (cds > 128u ? (uint64_t)tune("AEC_IDX_REGIONS_LONG", 3072) : 16u) || nreg > (1u << 22)) return p;
    p.nreg = (uint32_t)nreg;
    p.region_bits = region;
    p.avg_cds = (uint32_t)(rsi_bits_hint / c.rsi);
    // (the guess: an anchor -- two parses per bit of a coded data set --, the walk to an RSI start, tests, and the
    // candidate's RSI once more where RSIs are short; three-bit options say less: more of it)
    p.budget = tune("AEC_IDX_REGION_BUDGET", (uint32_t)((c.rsi <= kRgVerifyMaxRsi ? (c.id_len <= 3u ? 4u : 3u) : 2u) * c.rsi + 384u + 3u * cds));
    p.passes = tune("AEC_IDX_REGION_PASSES", 8);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    p.o_flags = o;  o = up(o + 64);
    p.o_found = o;  o = up(o + nreg * sizeof(RgEntry));
    for (int k = 0; k < 2; k++) {
        p.o_entry[k] = o; o = up(o + nreg * sizeof(RgEntry));
        p.o_exit[k] = o;  o = up(o + nreg * sizeof(RgState));
        p.o_cnt[k] = o;   o = up(o + nreg * 4);
    }
    p.o_base = o;   o = up(o + (nreg + 1) * 8);
    // the RSI starts a region's walk meets: room for four times what the hint expects (a walk takes the regions without
    // an entry behind it along) and a few
    p.K = (uint32_t)(4u * (region / rsi_bits_hint) + 8u);
    p.o_list = o;   o = up(o + nreg * p.K * 8);
    p.o_slist = o;
    if (want_segments) o = up(o + nreg * p.K * (size_t)c.segs_per_rsi * 8);
    p.bytes = o;
    p.ok = true;
    return p;
}

const uint32_t *launch_index_regions(const Cfg &c, const RegionPlan &p, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                                     uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t st,
                                     uint8_t *base, uint32_t start_block, uint64_t rsi_start, uint32_t tail_slot,
                                     uint64_t *d_seg_bits, const uint32_t *skip_if)
{
    const TrStream s{words, nwords, end_bit};
    RgTables t{};
    t.flags = reinterpret_cast<uint32_t *>(base + p.o_flags);
    t.found = reinterpret_cast<RgEntry *>(base + p.o_found);
    for (int k = 0; k < 2; k++) {
        t.entry[k] = reinterpret_cast<RgEntry *>(base + p.o_entry[k]);
        t.exit[k] = reinterpret_cast<RgState *>(base + p.o_exit[k]);
        t.cnt[k] = reinterpret_cast<uint32_t *>(base + p.o_cnt[k]);
    }
    t.base = reinterpret_cast<uint64_t *>(base + p.o_base);
    t.list = reinterpret_cast<uint64_t *>(base + p.o_list);
    t.slist = (d_seg_bits && p.o_slist != p.bytes) ? reinterpret_cast<uint64_t *>(base + p.o_slist) : nullptr;
    t.K = p.K;
    t.nreg = p.nreg;
    t.budget = p.budget;
    t.period = rg_ring_period(p.avg_cds);
    t.region_bits = p.region_bits;
    t.lo = start_bit;
    t.max_walk = 16u * p.region_bits;
    t.skip_if = skip_if;
    (void)hipMemsetAsync(t.flags, 0, 64, st);
    const uint32_t wg = (p.nreg + 63u) / 64u;
    const uint32_t gwg = wg < 2560u ? wg : 2560u;        // (the guesses come from a queue: as many wavefronts as the chip holds)
    const bool small_ring = p.avg_cds <= (uint32_t)tune("AEC_IDX_REGION_RING32", 32);
    hipLaunchKernelGGL(k_rg_guess, dim3(gwg), dim3(64), rg_lds_bytes(64u), st, c, s, t);
    hipLaunchKernelGGL(k_rg_link, dim3((p.nreg + 255u) / 256u), dim3(256), 0, st, t, start_bit, start_block);
    hipLaunchKernelGGL(k_rg_judge, dim3(1), dim3(1), 0, st, t);
    if (small_ring) hipLaunchKernelGGL(k_rg_walk<32u>, dim3(wg), dim3(64), rg_lds_bytes(32u), st, c, s, t);
    else hipLaunchKernelGGL(k_rg_walk<64u>, dim3(wg), dim3(64), rg_lds_bytes(64u), st, c, s, t);
    uint32_t cur = 0;
    for (uint32_t k = 0; k < p.passes; k++) {
        if (small_ring) hipLaunchKernelGGL(k_rg_mend<32u>, dim3(wg), dim3(64), rg_lds_bytes(32u), st, c, s, t, cur);
        else hipLaunchKernelGGL(k_rg_mend<64u>, dim3(wg), dim3(64), rg_lds_bytes(64u), st, c, s, t, cur);
        cur ^= 1u;
    }
    hipLaunchKernelGGL(k_rg_prep, dim3(1), dim3(1), 0, st, t);
    hipLaunchKernelGGL(k_rg_check, dim3((p.nreg + 255u) / 256u), dim3(256), 0, st, t, cur);
    hipLaunchKernelGGL(k_rg_sums, dim3(wg), dim3(64), 0, st, t, cur);
    hipLaunchKernelGGL(k_rg_scan, dim3(1), dim3(1024), 0, st, t, cur);
    hipLaunchKernelGGL(k_rg_fill, dim3(wg), dim3(64), rg_lds_bytes(64u), st, c, s, t, cur, d_rsi_off, max_rsi, d_res, tail_slot, rsi_start,
                       start_block, d_seg_bits);
#ifdef AEC_TUNING
    if (tune_set("AEC_IDX_STATS")) {                       // (diagnostics: synchronises)
        (void)hipStreamSynchronize(st);
        uint32_t fl[8] = {0};
        (void)hipMemcpy(fl, t.flags, 32, hipMemcpyDeviceToHost);
        std::vector<RgEntry> fo(p.nreg), en(p.nreg);
        std::vector<RgState> ex(p.nreg);
        (void)hipMemcpy(fo.data(), t.found, p.nreg * sizeof(RgEntry), hipMemcpyDeviceToHost);
        (void)hipMemcpy(en.data(), t.entry[0], p.nreg * sizeof(RgEntry), hipMemcpyDeviceToHost);
        (void)hipMemcpy(ex.data(), t.exit[0], p.nreg * sizeof(RgState), hipMemcpyDeviceToHost);
        uint32_t found = 0, mism = 0, lastlive = 0;
        for (uint32_t r = 1; r < p.nreg; r++) {
            found += fo[r].live;
            if (!en[r].live) continue;
            mism += ex[lastlive].st != 0u || ex[lastlive].pos != en[r].pos || ex[lastlive].b != en[r].b;
            lastlive = r;
        }
        fprintf(stderr, "regions: %u of %llu bits, budget %u | guesses found %u, kept %u, not the exit in front after the first walk "
                "%u, mended (all passes) %u | delivered %u, do not agree %u, input ended in region %u, stepped aside %u\n", p.nreg,
                (unsigned long long)p.region_bits, p.budget, found, fl[4], mism, fl[5], fl[0], fl[1], fl[2], fl[3]);
    }
#endif
    return t.flags;
}

}  // namespace aec
