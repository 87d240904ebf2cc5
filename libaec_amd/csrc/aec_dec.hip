// aec_dec.hip -- gfx950 decoder kernels for the CCSDS 121.0-B-2 adaptive entropy coder.
//
//   k_decode  ONE LANE PER RSI (or per 64-block SEGMENT when the encoder's segment table is at
//             hand: template parameter SEG).  A coded data set can only be located by parsing its
//             predecessor (the stream has no lengths, reference src/decode.c:402-421), so the
//             bit-serial dependency is kept inside a lane: the lane walks its RSI block by block
//             with a 64-bit window (reference decode.c:222-340), runs the inverse predictor
//             (decode.c:67-141), which is a serial chain as well, and stores whole blocks in the
//             caller's byte order (decode.c:144-189).  Parallelism comes from the RSIs: 4 GiB of
//             16-bit / block 16 / rsi 128 data is one million independent lanes.  The compressed
//             words reach a lane through its column of an LDS ring (16-byte loads issued ahead);
//             blocks of 16 or 32 bytes leave through LDS rows written out transposed, as whole
//             64-byte sectors -- per-lane 16-byte stores 2 KiB apart bounded the kernel.
//   (the index pass for streams that arrive without an offset table lives in aec_idx.hip)
//
// This file is compiled once per templated block size with -DAEC_DEC_PART=<0|8|16|32|64> (the kernels of that block size:
// objects aec_dec_bs<N>.o) and once without (everything that does not depend on the block size, and the dispatch): a
// process loads the code object of the block size it decodes, a fifth of the whole, with its first decode, and the five
// parts compile side by side.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "aec_kernels.h"
#include "aec_lane.h"
#include "aec_spec.h"
#include "aec_tune.h"

namespace aec {

// ---- decoding a bare stream segment by segment (launch_decode_bare) ---------------------------------------
// What the summing pass leaves per segment.  With u = (sample in front of the segment) - xmin:  the samples of the
// segment are u + (prefix sums of the steps) exactly when lo <= u <= hi, and then the sample behind the segment is
// u + sum.  (For the first segment of an RSI u is the reference sample `ref` itself and the sums start behind it.)
struct SegSum {
    int64_t sum, lo, hi;
    uint64_t end;          // bit behind the segment's last coded data set
    uint32_t ref;          // first segment of an RSI: the reference sample (raw)
    uint32_t ok;           // 1 = all blocks parsed, the segment ends on a coded data set
};

// what launch_decode_bare adds to a launch: the summing pass's output, or the list of RSIs to take
struct BareArgs {
    SegSum *sums = nullptr;
    const uint32_t *list = nullptr, *list_cnt = nullptr;
    uint64_t avg_hint = 0;         // bits per coded data set where the counts come from an index record
};

// The launch of one block size's kernels (defined in the object compiled with -DAEC_DEC_PART=BS, see the head of the file)
struct DecLaunch {
    const Cfg *c;
    const uint32_t *words;
    uint64_t nwords, end_bit;
    const uint64_t *rsi_off;
    const SegEntry *seg_table;
    uint64_t n_items, total_blocks;
    uint8_t *out;
    DecResult *res;
    hipStream_t st;
    uint8_t *dump;
    const DecResult *idx, *batch;
    uint32_t rpc;
    BareArgs ba;
};
template <int BS> void dec_part_bytes(bool seg, bool sums, const DecLaunch &a);      // k_decode<BS, ...>
template <int BS> void dec_part_wave(const DecLaunch &a);                             // k_decode_wave<BS, ...>

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

// Inverse predictor + byte-order store of one block (BS > 0: registers, vector stores).
//
// The predictor (reference decode.c:91-135): x += d / 2 or x -= (d + 1) / 2 as long as the step's size fits the room on
// both sides of x, something else where it does not.  Every sample of a block takes the first branch if the SUM of the
// block's mapped residuals fits the room on both sides of the sample in front of the block (a step is at most its
// residual, and x moves by at most the steps taken so far): one test per block, wave-wide, and then a sample is a
// shift, a sign, an exclusive or and an add (4.5 vector instructions with the sum, against 13 for the exact step; the
// predictor was a third of the kernel's 39 instructions per sample, profiles/r06/k_decode_isa.txt).  A wavefront with
// a lane whose block does not pass -- samples near the ends of the range -- takes the exact steps for that block.
template <int BS, int BYTES>
__device__ __forceinline__ void store_block(uint8_t *dst, const uint32_t *d, const Cfg &c, bool ref,
                                            uint32_t &x)
{
    const bool pp = c.flags & F_PREPROCESS, sgn = c.flags & F_SIGNED, msb = c.flags & F_MSB;
    uint32_t v[BS];
    if (!pp) {
#pragma unroll
        for (int i = 0; i < BS; i++) v[i] = d[i];
    } else {
        // (the first block of an RSI: its first sample is the reference sample itself, decode.c:78-86)
        const uint32_t x0 = ref ? (sgn ? sign_extend(d[0], c.bps) : d[0]) : x;
        const uint32_t d0 = ref ? 0u : d[0];
        uint32_t sum = d0, any = 0;
#pragma unroll
        for (int i = 1; i + 1 < BS; i += 2) sum += d[i] + d[i + 1];
        if (BS % 2 == 0) sum += d[BS - 1];
        if (BYTES == 4) {                          // (the sum of 64 residuals of up to 32 bits: none above 26 bits)
#pragma unroll
            for (int i = 0; i < BS; i++) any |= i == 0 ? d0 : d[i];
        }
        const uint32_t below = sgn ? x0 + c.xmax + 1u : x0, above = c.xmax - x0;      // x0 - xmin, xmax - x0
        const bool fits = sum <= below && sum <= above && (BYTES != 4 || (any >> 26) == 0u);
        if (!__any(!fits)) {
            uint32_t xx = x0;
#pragma unroll
            for (int i = 0; i < BS; i++) {
                const uint32_t di = i == 0 ? d0 : d[i];
                xx += (di >> 1) ^ (uint32_t)__builtin_amdgcn_sbfe((int)di, 0u, 1u);     // + d / 2, or - (d + 1) / 2
                v[i] = xx;
            }
            x = xx;
        } else {
#pragma unroll
            for (int i = 0; i < BS; i++) {
                if (i == 0 && ref) {
                    x = x0;
                } else {
                    x = sgn ? unpp_signed(x, d[i], c.xmax) : unpp_unsigned(x, d[i], c.xmax);
                }
                v[i] = x;
            }
        }
    }
    if (BYTES == 4) {
        uint4 *o = reinterpret_cast<uint4 *>(dst);
#pragma unroll
        for (int q = 0; q < BS / 4; q++) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = msb ? bswap32(v[4 * q + j]) : v[4 * q + j];
            o[q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else if (BYTES == 2) {
        // two samples per word, one byte permute each: low halves side by side, or their bytes swapped
        const uint32_t sel = msb ? 0x04050001u : 0x05040100u;
        uint32_t w[BS / 2];
#pragma unroll
        for (int j = 0; j < BS / 2; j++) w[j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], sel);
        if (BS >= 8) {
            uint4 *o = reinterpret_cast<uint4 *>(dst);
#pragma unroll
            for (int q = 0; q < BS / 8; q++) o[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    } else if (BYTES == 1) {
        // four samples per word: their low bytes, two permutes and an or
        uint32_t w[BS / 4];
#pragma unroll
        for (int j = 0; j < BS / 4; j++)
            w[j] = __builtin_amdgcn_perm(v[4 * j + 1], v[4 * j], 0x0c0c0400u) |
                   __builtin_amdgcn_perm(v[4 * j + 3], v[4 * j + 2], 0x04000c0cu);
        uint2 *o = reinterpret_cast<uint2 *>(dst);
#pragma unroll
        for (int q = 0; q < BS / 8; q++) o[q] = make_uint2(w[2 * q], w[2 * q + 1]);
    } else {   // 3-byte containers: byte stores
#pragma unroll
        for (int i = 0; i < BS; i++) {
            const uint32_t s = v[i];
            dst[3 * i + 0] = (uint8_t)(msb ? s >> 16 : s);
            dst[3 * i + 1] = (uint8_t)(s >> 8);
            dst[3 * i + 2] = (uint8_t)(msb ? s : s >> 16);
        }
    }
}

// generic block size / container: sample by sample
__device__ __forceinline__ void store_block_generic(uint8_t *dst, const uint32_t *d, const Cfg &c, bool ref,
                                                    uint32_t &x)
{
    const bool pp = c.flags & F_PREPROCESS, sgn = c.flags & F_SIGNED, msb = c.flags & F_MSB;
    for (uint32_t i = 0; i < c.bs; i++) {
        uint32_t v;
        if (!pp) v = d[i];
        else if (i == 0 && ref) v = x = sgn ? sign_extend(d[0], c.bps) : d[0];
        else v = x = sgn ? unpp_signed(x, d[i], c.xmax) : unpp_unsigned(x, d[i], c.xmax);
        for (uint32_t t = 0; t < c.bytes; t++)
            dst[i * c.bytes + t] = (uint8_t)(v >> (8 * (msb ? c.bytes - 1 - t : t)));
    }
}

__device__ __forceinline__ void report(DecResult *res, uint32_t status, uint64_t rsi, uint64_t block)
{
    // worst status wins (DATA_ERROR > NEED_INPUT > OK); remember the lowest failing RSI and -- in the decode record's
    // tail_blocks, which the decoder has no other use for -- the lowest failing block of the batch: everything in
    // front of it is decoded and valid (reference: every sample preceding an error is delivered)
    atomicMax(&res->status, status);
    atomicMin(reinterpret_cast<unsigned long long *>(&res->bad_rsi), (unsigned long long)rsi);
    atomicMin(reinterpret_cast<unsigned long long *>(&res->tail_blocks), (unsigned long long)block);
}

// ---- compressed-stream staging ------------------------------------------------------------------
// Every lane owns one column of a small LDS ring: word i (relative to the lane's 16-byte aligned
// base a0) lives at ring[((slot0 + i) & mask) * 64 + lane], so a wave-wide access at any per-lane
// index is bank-conflict free.  The ring is topped up at ONE wave-synchronous point per block
// iteration with 16-byte global loads issued an iteration ahead (issue early / write late), so the
// bit reader itself only ever touches LDS and never waits on HBM.
struct RingSrc {
    const uint32_t *col;   // ring + lane
    uint32_t slot0, mask;
    uint32_t limit;        // words landed (relative index); reads at or beyond it starve
    uint32_t starve;
    __device__ __forceinline__ uint32_t word(uint32_t i)
    {
        starve |= (i >= limit) ? 1u : 0u;      // the slot read is always inside the ring
        return col[((slot0 + i) & mask) * 64u];
    }
    // words i and i+1; row mask+1 mirrors row 0 so the pair is always two adjacent rows
    __device__ __forceinline__ void word2(uint32_t i, uint32_t &w0, uint32_t &w1)
    {
        starve |= (i + 1 >= limit) ? 1u : 0u;
        const uint32_t *q = col + ((slot0 + i) & mask) * 64u;
        w0 = q[0];
        w1 = q[64];
    }
    // words i, i+1, i+2 (rows mask+1 and mask+2 mirror rows 0 and 1)
    __device__ __forceinline__ void word3(uint32_t i, uint32_t &w0, uint32_t &w1, uint32_t &w2)
    {
        starve |= (i + 2 >= limit) ? 1u : 0u;
        const uint32_t *q = col + ((slot0 + i) & mask) * 64u;
        w0 = q[0];
        w1 = q[64];
        w2 = q[128];
    }
    __device__ __forceinline__ bool starved() const { return starve != 0; }
};

// four consecutive stream words starting at absolute index idx (multiple of 4), RAW byte order
// (the swap to host order happens when they are written to the ring, so that a load issued one
// iteration ahead is not waited for at issue time); zeros past the end of the buffer
__device__ __forceinline__ uint4 load_words4(const uint32_t *__restrict__ words, uint64_t idx, uint64_t nwords)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (idx + 4 <= nwords) {
        v = *reinterpret_cast<const uint4 *>(words + idx);
    } else {
        if (idx < nwords) v.x = words[idx];
        if (idx + 1 < nwords) v.y = words[idx + 1];
        if (idx + 2 < nwords) v.z = words[idx + 2];
    }
    return v;
}

// branch-free variant for the steady-state prefetch: idx is clamped to the last whole 16-byte chunk
// of the buffer (the words past the end of a stream are never consumed by a valid CDS; the rare
// exact tail is served by the synchronous refill with load_words4)
__device__ __forceinline__ uint4 load_words4_nb(const uint32_t *__restrict__ words, uint64_t idx, uint64_t last_chunk)
{
    if (idx > last_chunk) idx = last_chunk;
    return *reinterpret_cast<const uint4 *>(words + idx);
}

// rel is a multiple of 4 and slot0 a multiple of 4, so the four rows never wrap inside a chunk
__device__ __forceinline__ void ring_put4(uint32_t *col, uint32_t slot0, uint32_t mask, uint32_t rel, uint4 v)
{
    const uint32_t row = (slot0 + rel) & mask;
    uint32_t *q = col + row * 64u;
    const uint32_t x = bswap32(v.x), y = bswap32(v.y);
    q[0] = x;
    q[64] = y;
    q[128] = bswap32(v.z);
    q[192] = bswap32(v.w);
    if (row == 0) {                               // mirrors of rows 0 and 1 for word2() / word3()
        col[(mask + 1u) * 64u] = x;
        col[(mask + 2u) * 64u] = y;
    }
}

// accumulator of the summing pass: 32 bits carry a segment of up to 4096 samples of at most 16 bits
template <int BYTES>
struct SumAcc {
    typedef int64_t type;
    static constexpr int64_t kLo = -((int64_t)1 << 60), kHi = (int64_t)1 << 60;
};
template <>
struct SumAcc<1> {
    typedef int32_t type;
    static constexpr int32_t kLo = -(1 << 30), kHi = 1 << 30;
};
template <>
struct SumAcc<2> {
    typedef int32_t type;
    static constexpr int32_t kLo = -(1 << 30), kHi = 1 << 30;
};

// one block of mapped residuals d[] into the running sum P and the bounds (reference decode.c:96-134: the step is
// two-sided, +d/2 or -(d+1)/2, while half = ceil(d/2) <= min(x - xmin, xmax - x))
template <int BS, typename ACC>
__device__ __forceinline__ void seg_accumulate(const uint32_t *d, bool first_is_ref, ACC R, ACC &P, ACC &lo, ACC &hi)
{
#pragma unroll
    for (int i = 0; i < BS; i++) {
        const uint32_t v = (i == 0 && first_is_ref) ? 0u : d[i];     // (the reference sample is no step)
        const ACC h = (ACC)((v >> 1) + (v & 1u));
        const ACC step = (v & 1u) ? -h : h;
        const ACC nlo = h - P, nhi = R - h - P;
        lo = nlo > lo ? nlo : lo;
        hi = nhi < hi ? nhi : hi;
        P += step;
    }
}

// Build-time knobs of k_decode for A/B runs (tests/ab_build.sh builds a variant library, AEC_AMD_LIB selects it);
// the defaults are the measured best:
//   AEC_STG_ROW / AEC_STG_ROW8   bytes per output staging row for 16- / 32-byte blocks and for 8-byte blocks (0 = none)
//   AEC_DEC_UNR8 / AEC_DEC_UNR16 blocks of 8 / 16 samples decoded per top-up of the ring
//   AEC_DEC_OU                   1 = the loop body covers a whole staging row (counted vmcnt past the row stores)
//   AEC_DEC_MINW                 second __launch_bounds__ argument (waves per SIMD the register allocator must allow)
#ifndef AEC_STG_ROW
#define AEC_STG_ROW 64
#endif
#ifndef AEC_STG_ROW8
#define AEC_STG_ROW8 32
#endif
#ifndef AEC_DEC_MINW
#define AEC_DEC_MINW 1
#endif
#ifndef AEC_DEC_UNR8
#define AEC_DEC_UNR8 2
#endif
#ifndef AEC_DEC_OU
#define AEC_DEC_OU 1
#endif
#ifndef AEC_DEC_UNR16
#define AEC_DEC_UNR16 1
#endif
__host__ __device__ constexpr uint32_t dec_unroll(int bs) { return bs == 8 ? AEC_DEC_UNR8 : (bs == 16 ? AEC_DEC_UNR16 : 1); }
__host__ __device__ constexpr bool stg_on(int blk) { return blk == 16 || blk == 32 || (blk == 8 && AEC_STG_ROW8 != 0); }
__host__ __device__ constexpr uint32_t stg_row(int blk) { return blk == 8 ? (AEC_STG_ROW8 ? AEC_STG_ROW8 : 64) : AEC_STG_ROW; }

// SEG = false: work item = RSI, start bits from rsi_off.  SEG = true: work item = segment (64
// blocks), start bit and preceding sample from the encoder's segment table.
// kPend = 16-byte loads kept in flight per lane across one block iteration: 2 feed 256 bits per block, enough
// for the short coded data sets of compressible data; streams that average more per block (large blocks,
// high-entropy data: typical.dat's 64-sample blocks at 720 bits) would drain the ring and fall into the
// synchronous refill -- an HBM round trip per 16 bytes -- every iteration, so they run with 4 or 8.
//
// SUMS (with SEG): the first of the two passes that decode a BARE stream segment by segment (launch_decode_bare).
// The items are segments whose start bits the index pass found (rsi_off[item], ~0 = unknown) -- but the sample in
// front of a segment is what only decoding gives.  Inside the range the inverse predictor is a running sum
// (reference decode.c:96-134: x += d / 2 or x -= (d + 1) / 2 as long as the step stays within the room on both
// sides), so a lane parses its segment, sums the steps and records for which predecessors the running sum IS the
// predictor (SegSum: lo <= predecessor - xmin <= hi); nothing is written to `out` or `res`.  k_seg_scan then
// chains the sums along each RSI, and the second pass is this kernel with SEG alone.
template <int BS, int BYTES, bool SEG, int kPend, bool SUMS = false>
__global__ void __launch_bounds__(256, AEC_DEC_MINW)
k_decode(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
         const uint64_t *__restrict__ rsi_off, const SegEntry *__restrict__ seg_table, uint64_t n_rsi,
         uint64_t total_blocks, uint8_t *__restrict__ out, DecResult *res, uint32_t ring_words, uint32_t maxw,
         uint32_t needw, uint8_t *__restrict__ dump, const DecResult *__restrict__ idx,
         const DecResult *__restrict__ batch, uint32_t rsi_per_chunk, SegSum *__restrict__ sums,
         const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_cnt)
{
    static_assert(!SUMS || (SEG && BS != 0), "the summing pass runs per segment on the templated block sizes");
    // item counts straight from the record the index pass left on the device (the grid was sized for
    // the most it could find): no host round trip between the two passes
    if (idx) {
        const uint64_t whole = idx->n_rsi, tail = idx->tail_blocks;
        n_rsi = whole + (tail ? 1u : 0u);
        total_blocks = whole * c.rsi + tail;
        if (SEG) n_rsi *= c.segs_per_rsi;             // (items are segments)
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t nwords_vec = nwords >= 4 ? ((nwords + 3) & ~3ull) - 4 : 0;   // last 16-byte chunk
    // Output staging (small blocks): a lane produces BLK = 8..32 bytes per iteration, 2 KiB away from
    // its neighbours' -- stored directly, every 16-byte store of a wave touches 64 different lines
    // and the L2 sees partial-sector writes only; that store path, not decoding, bounded the kernel
    // (TA stalled by TC 77 % of the time; 3.6 ms -> 2.2 ms with the stores pointed at one line).
    // So blocks are parked in an LDS row per lane and every kStgRow bytes the wave writes the rows
    // out transposed: 4 lanes per row, whole 64-byte sectors per store instruction.  (Rows of 128
    // bytes -- whole lines -- were slower: 3.26 ms against 2.75 ms at C2, the LDS costs waves.)
    constexpr int BLK = BS * BYTES;
    constexpr bool STG = !SUMS && stg_on(BLK);
    constexpr uint32_t kStgRow = stg_row(BLK), kStgStride = kStgRow + 16u;   // 16 bytes of padding: conflict-free rows
    constexpr uint32_t G = STG ? kStgRow / (uint32_t)(BLK ? BLK : 1) : 1u;
    const uint32_t wave_words = (ring_words + 2u) * 64u + (STG ? (64u * kStgStride) / 4u : 0u);
    uint32_t *wbase = smem + (size_t)wave * wave_words;
    uint32_t *col = wbase + lane;
    uint8_t *stage = reinterpret_cast<uint8_t *>(wbase + (ring_words + 2u) * 64u);
    const uint32_t mask = ring_words - 1;

    uint64_t r = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 64u + lane;
    bool active = r < n_rsi;
    if (!SEG && list) {                               // the items are the RSIs of a list (launch_decode_bare: the
        active = r < *list_cnt;                       // RSIs that could not be taken segment by segment)
        r = active ? list[r] : 0u;
    }
    constexpr int DN = BS ? BS : (int)kMaxBlockSize;
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const bool pp = c.flags & F_PREPROCESS;

    uint32_t nb = 0, b0 = 0, x = 0;
    uint64_t start = 0, first_blk = 0;
    if (active) {
        if (SEG) {
            const uint64_t rsi_idx = (r >> 32) ? r / c.segs_per_rsi : (uint64_t)((uint32_t)r / c.segs_per_rsi);
            b0 = (uint32_t)(r - rsi_idx * c.segs_per_rsi) * 64u;
            uint64_t left = total_blocks > rsi_idx * c.rsi ? total_blocks - rsi_idx * c.rsi : 0u;
            if (left > c.rsi) left = c.rsi;
            nb = left <= b0 ? 0u : (left - b0 > 64 ? 64u : (uint32_t)(left - b0));
            if (SUMS) {
                start = rsi_off[r];
            } else {
                const SegEntry e = seg_table[r];
                start = e.bit;
                x = (c.flags & F_SIGNED) ? sign_extend(e.prev, c.bps) : e.prev;
            }
            if (start == ~0ull) {                     // (a segment the table does not know: not this pass's)
                start = 0;
                nb = 0;
            }
            first_blk = rsi_idx * c.rsi + b0;
        } else if (batch) {
            // batch of independent streams: stream s owns RSIs [s * rsi_per_chunk, (s + 1) * rsi_per_chunk) of
            // the table and of the output; what its index pass found says how many of them hold blocks
            const uint64_t sidx = r / rsi_per_chunk;
            const uint32_t rin = (uint32_t)(r - sidx * rsi_per_chunk);
            const uint64_t whole = batch[sidx].n_rsi, tail = batch[sidx].tail_blocks;
            nb = rin < whole ? c.rsi : (rin == whole ? (uint32_t)tail : 0u);
            start = nb ? rsi_off[r] : 0;
            first_blk = r * c.rsi;
        } else {
            uint64_t left = total_blocks - r * c.rsi;
            nb = left > c.rsi ? c.rsi : (uint32_t)left;
            start = rsi_off[r];
            first_blk = r * c.rsi;
        }
    }
    const size_t blk_bytes = (size_t)bs * c.bytes;
    uint8_t *dst = out + (size_t)first_blk * blk_bytes;
    // (a row's destination and fill level live in its lane's registers; the lanes that write the row out
    // fetch them with a lane permute -- as pointer and count arrays they cost 768 bytes of LDS per wave, the
    // 16th wave of a CU at C2)
    const uint32_t dst_lo = (uint32_t)reinterpret_cast<uintptr_t>(dst), dst_hi = (uint32_t)(reinterpret_cast<uintptr_t>(dst) >> 32);
    uint32_t produced = 0;                              // blocks this lane parked in the current group
    auto flush = [&](uint32_t group) {
        const uint32_t filled = produced * (uint32_t)BLK;
        produced = 0;
        constexpr uint32_t LPR = kStgRow / 16u, RPI = 64u / LPR;   // lanes per row, rows per store instruction
#pragma unroll
        for (uint32_t k = 0; k < LPR; k++) {
            const uint32_t row = k * RPI + lane / LPR, chunk = (lane % LPR) * 16u;
            const uint4 v = *reinterpret_cast<const uint4 *>(stage + row * kStgStride + chunk);
            const uint32_t have = __shfl(filled, (int)row);
            const uintptr_t rbase = (uintptr_t)__shfl(dst_lo, (int)row) | ((uintptr_t)__shfl(dst_hi, (int)row) << 32);
            uint8_t *at = reinterpret_cast<uint8_t *>(rbase) + (size_t)group * kStgRow + chunk;
            uint8_t *q = chunk + 16u <= have ? at : dump + (size_t)lane * 16u;
            // (the row's base pointer comes out of LDS: say that it points to global memory, or
            // the store is emitted as a flat instruction)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) u32x4 global_u32x4;
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            // Whole 64-byte rows are written once and never read here: non-temporal, so that they do not
            // push the stream lines the lanes are still reading out of the L2 (2.50 -> 2.40 ms at C2).  The
            // 32-byte rows of 8-byte blocks need the L2 to merge them into sectors: 4.8 -> 6.5 ms with nt.
            if (kStgRow >= 64u)
                __builtin_nontemporal_store(vv, reinterpret_cast<global_u32x4 *>(reinterpret_cast<uintptr_t>(q)));
            else
                *reinterpret_cast<global_u32x4 *>(reinterpret_cast<uintptr_t>(q)) = vv;
            if (BLK == 8) {                          // an odd number of 8-byte blocks ends in half a chunk
                const bool half = have == chunk + 8u;
                if (__any(half)) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(1))) u32x2 global_u32x2;
                    const u32x2 hv = {v.x, v.y};
                    if (half) *reinterpret_cast<global_u32x2 *>(reinterpret_cast<uintptr_t>(at)) = hv;
                }
            }
        }
    };

    const uint64_t a0 = (start >> 5) & ~3ull;        // lane base word, 16-byte aligned
    const uint32_t slot0 = (uint32_t)a0 & mask;
    uint32_t landed = 0;
    for (; landed < ring_words; landed += 4)          // prologue: fill the whole ring
        ring_put4(col, slot0, mask, landed, load_words4(words, a0 + landed, nwords));

    RingSrc src{col, slot0, mask, landed, 0u};
    const uint64_t left_bits = end_bit - a0 * 32u;
    const uint32_t end_p = left_bits > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)left_bits;
    uint32_t p = (uint32_t)(start - a0 * 32u);        // fast path: bit position relative to a0
    WinReader<RingSrc> br;                            // generic path: sequential reader
    if (BS == 0) br.init(src, a0 * 32u, end_bit, p);

    uint4 pend[kPend];
    uint32_t pv[kPend];                                   // (0 / 1 in a VGPR: lane masks carried around the loop cost SGPR pairs)
#pragma unroll
    for (int j = 0; j < kPend; j++) { pend[j] = make_uint4(0, 0, 0, 0); pv[j] = 0u; }

    uint32_t d[DN];
    uint32_t zrun = 0;
    uint32_t ok = 1u;                                     // (a VGPR, like pv[])
    // SUMS: running sum of the predictor steps, and the bounds on (predecessor - xmin) inside which it is exact
    typedef typename SumAcc<BYTES>::type acc_t;
    const acc_t acc_R = (acc_t)(((uint64_t)1 << c.bps) - 1u);
    acc_t acc_P = 0, acc_lo = SumAcc<BYTES>::kLo, acc_hi = SumAcc<BYTES>::kHi;
    uint32_t sum_ref = 0, sum_bad = 0, sum_done = 0;

    // Blocks of 8 samples: UNR = 2 of them per loop iteration -- the ring top-up, the landing of the loads
    // in flight and the issue of the next ones are paid once per 16 samples like for the larger blocks.
    constexpr uint32_t UNR = (STG && G % dec_unroll(BS) != 0) ? 1u : dec_unroll(BS);   // (a row holds whole top-up groups)
    // The loop body covers one whole staging group (OU top-ups of UNR blocks = the G blocks of a row), so
    // the flush at its end is straight-line code: the wait that lands the loads in flight can then be
    // counted past the flush's stores (vmcnt(n), n = the stores behind the loads) instead of draining
    // them -- with the flush under a condition the compiler has to wait for the stores' completion, and
    // a store to HBM takes longer than a block decode (2.59 -> 2.14 ms at C2 with the stores sent to one
    // cached line: what the drain cost).
    constexpr uint32_t OU = (STG && AEC_DEC_OU && G > UNR) ? G / UNR : 1u;
    static_assert(!STG || G % UNR == 0, "a staging row holds whole top-up groups");
    uint32_t b = 0;
    for (; __any(b < nb && ok); b += OU * UNR) {
#pragma unroll
      for (uint32_t ou = 0; ou < OU; ou++) {
        const uint32_t bo = b + ou * UNR;
        const bool live = bo < nb && ok;
        // Order inside one iteration: (rare) synchronous refill -> decode -> land the 16-byte loads
        // issued one iteration ago -> store the block -> issue the next loads.  At the landing
        // point everything still outstanding (those loads, the previous block's stores) was
        // issued at least one whole block decode earlier, so the vmcnt wait there is free.
        {
            const uint32_t next = BS ? (p >> 5) : br.next;         // first word still needed
            if (__any(live && landed - next < needw)) {            // rare: a lane fell behind
#pragma unroll
                for (int j = 0; j < kPend; j++) {                  // in-flight chunks come first
                    if (pv[j]) {
                        ring_put4(col, slot0, mask, landed, pend[j]);
                        landed += 4;
                    }
                    pv[j] = 0u;
                }
                while (__any(live && landed - next < needw)) {
                    uint4 t[kPend];                                // a round of loads per wait, not one
#pragma unroll
                    for (int j = 0; j < kPend; j++) t[j] = load_words4(words, a0 + landed + 4u * j, nwords);
#pragma unroll
                    for (int j = 0; j < kPend; j++)
                        if (live && landed - next < needw && landed + 4u - next <= ring_words) {
                            ring_put4(col, slot0, mask, landed, t[j]);
                            landed += 4;
                        }
                }
            }
        }

        // ---- one block per lane ----
        const uint32_t ref = (pp && b0 + bo == 0) ? 1u : 0u;       // per lane in SEG mode
        if (BS) {
#pragma unroll
          for (uint32_t uu = 0; uu < UNR; uu++) {
            const uint32_t bb = bo + uu;
            const bool live = bb < nb && ok;
            const uint32_t ref = (pp && b0 + bb == 0) ? 1u : 0u;
            src.limit = landed;
            src.starve = 0u;
            uint32_t nz = 0;
            const bool parse = live && zrun == 0;
            // lanes inside a zero run (and finished lanes) get d = 0 from the same code path
            const uint32_t p_save = p;
            uint32_t st;
#pragma nounroll
            for (uint32_t attempt = 0;; attempt++) {
                st = decode_block<(BS ? BS : 2)>(src, p, end_p, d, c, ref, b0 + bb, parse, nz);
                if (attempt != 0 || needw >= maxw || !__any(parse && src.starved())) break;
                // Half-size ring (see dec_geom): a CDS longer than the look-ahead kept in steady
                // state.  Land what is in flight, fill the ring completely from this block's first
                // word -- that covers any CDS -- and decode the block again (all lanes, same result
                // for those that had enough).
                const uint32_t first = p_save >> 5;
#pragma unroll
                for (int j = 0; j < kPend; j++) {
                    if (pv[j]) {
                        ring_put4(col, slot0, mask, landed, pend[j]);
                        landed += 4;
                    }
                    pv[j] = 0u;
                }
                while (__any(live && landed + 4u - first <= ring_words)) {
                    if (live && landed + 4u - first <= ring_words) {
                        ring_put4(col, slot0, mask, landed, load_words4(words, a0 + landed, nwords));
                        landed += 4;
                    }
                }
                src.limit = landed;
                src.starve = 0;
                p = p_save;
                nz = 0;
            }
            // The block took bits the ring never held: a coded data set longer than any the reference encoder
            // writes (the ring is sized for those; the format itself knows no bound).  Whatever came out is
            // void, status included: the item is decoded again from the stream itself (k_decode_redo).
            // (a status raised while reads went beyond the ring -- stale slots look like a cut or corrupt stream --
            // is no verdict either)
            const bool over = parse && (((p + 31u) >> 5) > landed || (st != DEC_OK && src.starved()));
            if (SUMS) {                                   // (no verdicts here: such a segment's RSI takes the other path)
                if (parse && (over || st != DEC_OK)) {
                    sum_bad = 1u;
                    ok = 0u;
                } else if (parse && nz) {
                    zrun = nz;
                }
            } else if (over) {
                atomicOr(&res->pad, kDecRedo);
                ok = 0u;
            } else if (parse) {
                if (st != DEC_OK) {
                    report(res, st, SEG ? (first_blk + bb) / c.rsi : r, first_blk + bb);
                    // (a batch of independent streams: the stream's own record says so as well -- one overall
                    // record names only the first bad RSI of the whole batch)
                    if (batch && st == DEC_DATA_ERROR)
                        atomicMax(&const_cast<DecResult *>(batch)[r / rsi_per_chunk].status, (uint32_t)DEC_DATA_ERROR);
                    ok = 0u;
                } else if (nz) {
                    zrun = nz;
                }
            }
            // VMEM order per iteration: land the loads issued one iteration ago, issue the next
            // loads, then store the block.  Every lane issues every instruction (idle lanes re-read
            // their base chunk / write to a dump slot), so the instruction count per iteration is
            // fixed and the wait at the landing point can leave the younger stores outstanding.
            if (uu == 0) {
#pragma unroll
                for (int j = 0; j < kPend; j++)
                    if (pv[j]) {
                        ring_put4(col, slot0, mask, landed, pend[j]);
                        landed += 4;
                    }
                const uint32_t next = p >> 5;
#pragma unroll
                for (int j = 0; j < kPend; j++) {
                    pv[j] = (live && ok && (landed + 4u * j + 4u - next <= ring_words)) ? 1u : 0u;
                    const uint64_t idx = pv[j] ? a0 + landed + 4u * j : 0;   // idle lanes share one line
                    pend[j] = load_words4_nb(words, idx, nwords_vec);
                }
            }
            if (SUMS) {
                const bool st_ok = live && ok;
                const bool rf = ref != 0 && parse;
                if (st_ok) {
                    if (rf) sum_ref = d[0];
                    if (c.flags & F_PREPROCESS) seg_accumulate<DN, acc_t>(d, rf, acc_R, acc_P, acc_lo, acc_hi);
                    sum_done++;
                }
                zrun -= (st_ok && zrun) ? 1u : 0u;
            } else if (STG) {
                const bool st_ok = live && ok;
                store_block<DN, (BYTES ? BYTES : 1)>(stage + lane * kStgStride + (bb % G) * (uint32_t)BLK, d, c,
                                                     ref != 0 && parse, x);
                produced += st_ok ? 1u : 0u;
                zrun -= (st_ok && zrun) ? 1u : 0u;
                // (b is a multiple of the group, so the position inside it is a compile-time constant)
                if (OU * UNR == G ? (ou * UNR + uu == G - 1u) : ((bb % G) == G - 1u)) flush(bb / G);
            } else {
                const bool st_ok = live && ok;
                uint8_t *q = st_ok ? dst : dump + (size_t)lane * blk_bytes;
                store_block<DN, (BYTES ? BYTES : 1)>(q, d, c, ref != 0 && parse, x);
                dst += st_ok ? blk_bytes : 0;
                zrun -= (st_ok && zrun) ? 1u : 0u;
            }
          }
        } else if (live) {
            br.src.limit = landed;
            bool rf = ref != 0;
            if (zrun == 0) {
                uint32_t nz = 0;
                const uint32_t st = parse_cds<0>(br, d, c, ref, b0 + bo, nz);
                if (br.src.starved()) {                           // (as above: not a verdict on the stream)
                    atomicOr(&res->pad, kDecRedo);
                    ok = 0u;
                } else if (st != DEC_OK) {
                    report(res, st, r, first_blk + bo);
                    // (a batch of independent streams: the stream's own record says so as well -- one overall
                    // record names only the first bad RSI of the whole batch)
                    if (batch && st == DEC_DATA_ERROR)
                        atomicMax(&const_cast<DecResult *>(batch)[r / rsi_per_chunk].status, (uint32_t)DEC_DATA_ERROR);
                    ok = 0u;
                } else if (nz) {
                    const uint32_t keep = d[0];
                    for (int i = 0; i < DN; i++) d[i] = 0;
                    if (rf) d[0] = keep;
                    zrun = nz - 1;
                }
            } else {
                zrun--;
                rf = false;
            }
            if (ok) {
                store_block_generic(dst, d, c, rf, x);
                dst += blk_bytes;
                if (rf) d[0] = 0;
            }
        }
    }
    }
    if (STG && (b % G) != 0) flush(b / G);             // rows of the last, partial group
    if (SUMS) {
        // good: every block of the segment parsed, and no zero run reaches beyond it (then the next segment does
        // not start on a coded data set and the RSI is not one to take segment by segment)
        if (active) {
            SegSum o;
            o.sum = (int64_t)acc_P;
            o.lo = (int64_t)acc_lo;
            o.hi = (int64_t)acc_hi;
            o.end = a0 * 32u + p;
            o.ref = sum_ref;
            o.ok = (nb != 0u && !sum_bad && sum_done == nb && zrun == 0u) ? 1u : 0u;
            sums[r] = o;
        }
        return;
    }
    // The predictor state behind the LAST item of the batch, as the reference carries it (32 bits, not cut to the
    // sample width: on damaged streams it leaves the range): k_decode_partial continues from it.
    if (!SEG && active && (list ? first_blk + nb == total_blocks : r + 1 == n_rsi)) res->end_bit = x;
    if (SEG && active && nb && first_blk + nb == total_blocks) res->end_bit = x;
}

// ---- coded data sets longer than the ring ------------------------------------------------------------
// k_decode flags the batch (kDecRedo in the record's pad) when a lane met a coded data set that outgrew its
// ring.  This kernel is enqueued behind every k_decode and returns at once unless the flag is up; then it
// decodes EVERY item of the batch again, one lane per item, bit by bit from the stream where it lies
// (BitReader + parse_cds, the sequential reader the index pass and the emulator use): slow, but the answer for
// any stream the format allows -- a foreign encoder that picks k = 0 for large residuals is within its rights.
// Items and results as in k_decode (same tables, same output, same status record).
template <bool SEG>
__global__ void __launch_bounds__(64)
k_decode_redo(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
              const uint64_t *__restrict__ rsi_off, const SegEntry *__restrict__ seg_table, uint64_t n_rsi,
              uint64_t total_blocks, uint8_t *__restrict__ out, DecResult *res, const DecResult *__restrict__ idx,
              const DecResult *__restrict__ batch, uint32_t rsi_per_chunk, const uint32_t *__restrict__ list,
              const uint32_t *__restrict__ list_cnt)
{
    if (!(*reinterpret_cast<volatile uint32_t *>(&res->pad) & kDecRedo)) return;
    if (idx) {
        const uint64_t whole = idx->n_rsi, tail = idx->tail_blocks;
        n_rsi = whole + (tail ? 1u : 0u);
        total_blocks = whole * c.rsi + tail;
        if (SEG) n_rsi *= c.segs_per_rsi;
    }
    if (!SEG && list) n_rsi = *list_cnt;
    const bool pp = c.flags & F_PREPROCESS;
    const size_t blk_bytes = (size_t)c.bs * c.bytes;
    for (uint64_t r_lin = (uint64_t)blockIdx.x * 64u + threadIdx.x; r_lin < n_rsi; r_lin += (uint64_t)gridDim.x * 64u) {
        const uint64_t r = (!SEG && list) ? list[r_lin] : r_lin;
        uint32_t nb = 0, b0 = 0, x = 0;
        uint64_t start = 0, first_blk = 0;
        if (SEG) {
            const uint64_t rsi_idx = r / c.segs_per_rsi;
            b0 = (uint32_t)(r - rsi_idx * c.segs_per_rsi) * 64u;
            uint64_t left = total_blocks > rsi_idx * c.rsi ? total_blocks - rsi_idx * c.rsi : 0u;
            if (left > c.rsi) left = c.rsi;
            nb = left <= b0 ? 0u : (left - b0 > 64 ? 64u : (uint32_t)(left - b0));
            const SegEntry e = seg_table[r];
            if (e.bit == ~0ull) continue;                         // (not an item of this batch: launch_decode_bare)
            start = e.bit;
            x = (c.flags & F_SIGNED) ? sign_extend(e.prev, c.bps) : e.prev;
            first_blk = rsi_idx * c.rsi + b0;
        } else if (batch) {
            const uint64_t sidx = r / rsi_per_chunk;
            const uint32_t rin = (uint32_t)(r - sidx * rsi_per_chunk);
            const uint64_t whole = batch[sidx].n_rsi, tail = batch[sidx].tail_blocks;
            nb = rin < whole ? c.rsi : (rin == whole ? (uint32_t)tail : 0u);
            start = nb ? rsi_off[r] : 0;
            first_blk = r * c.rsi;
        } else {
            const uint64_t left = total_blocks - r * c.rsi;
            nb = left > c.rsi ? c.rsi : (uint32_t)left;
            start = rsi_off[r];
            first_blk = r * c.rsi;
        }
        uint8_t *dst = out + (size_t)first_blk * blk_bytes;
        BitReader br;
        br.init(words, nwords, end_bit, start);
        uint32_t d[kMaxBlockSize];
        uint32_t zrun = 0;
        for (uint32_t bo = 0; bo < nb; bo++) {                    // (the sample-by-sample path of k_decode)
            bool rf = pp && b0 + bo == 0;
            if (zrun == 0) {
                uint32_t nz = 0;
                const uint32_t st = parse_cds<0>(br, d, c, rf ? 1u : 0u, b0 + bo, nz);
                if (st != DEC_OK) {
                    report(res, st, r, first_blk + bo);
                    if (batch && st == DEC_DATA_ERROR)
                        atomicMax(&const_cast<DecResult *>(batch)[r / rsi_per_chunk].status, (uint32_t)DEC_DATA_ERROR);
                    break;
                }
                if (nz) {
                    const uint32_t keep = d[0];
                    for (uint32_t i = 0; i < kMaxBlockSize; i++) d[i] = 0;
                    if (rf) d[0] = keep;
                    zrun = nz - 1;
                }
            } else {
                zrun--;
                rf = false;
            }
            store_block_generic(dst, d, c, rf, x);
            dst += blk_bytes;
            if (rf) d[0] = 0;
        }
        // (as k_decode: the state k_decode_partial continues from)
        if (!SEG && (list ? first_blk + nb == total_blocks : r + 1 == n_rsi)) res->end_bit = x;
        if (SEG && nb && first_blk + nb == total_blocks) res->end_bit = x;
    }
}

// ---- small and medium streams: ONE WAVEFRONT PER RSI, ONE LANE PER BLOCK (round 5) -------------------------------
// k_decode gives an RSI to a lane: its 128 blocks are a chain of ~1.5 us each when nothing else hides the latency, so a
// 64 KiB chunk, a 1 MiB chunk and a 16 MiB stream all took the same 197 us while 99 % of the chip idled.  Here a
// wavefront owns the RSI.  Two phases per round of up to 64 blocks:
//   walk    the coded data sets are LOCATED one behind the other (their lengths only: reference src/decode.c:402-421 has
//           no other way in).  A wave-cooperative parse per coded data set (the index walker's: masked popcounts, DPP
//           prefix sum, ballot, rank select) costs ~0.8 us each -- cross-lane round trips one behind the other -- and
//           made this kernel 105 us for an RSI of 128 blocks.  So the 64 lanes parse the coded data sets that WOULD
//           begin at the next 64 BITS, one each: ~45 vector instructions and no dependence between the lanes, because
//           the end of a unary part is one lookup once the positions of the 1-bits of a piece of the stream (4096
//           bits) lie in a table (n-th 1-bit behind q = ones[rank(q) + n - 1]).  The chain through those 64 bits is
//           then one lane read per coded data set.  (The same lengths in an LDS table for a whole piece, then one LDS
//           read per step: 320 cycles per coded data set against ~180, and 8 KB more per wavefront.)  Only the first
//           coded data set of the RSI (the one with the reference sample), zero runs and whatever the table does not
//           resolve take the cooperative parse.  The start of block i stays in lane i's register.
//   decode  every lane decodes ITS block (decode_block, the same code k_decode runs per lane); the inverse predictor,
//           a serial chain over the samples in the reference (decode.c:96-134), runs per block from a GUESSED
//           predecessor -- the running sum of the steps, exact whenever nothing clips at the ends of the range -- and
//           the guesses are checked: lane i's input must be lane i-1's output; the first lane where it is not takes
//           the true value, the lanes behind it shift, and the blocks are computed again until all agree (one pass on
//           data that stays inside the range, one more per clipping block otherwise: exactness never rests on the guess)
//   store   a lane's block lies behind its neighbour's: whole lines per store instruction, no staging
// Items, tables, result record and error reporting as k_decode<SEG = false> (no list mode); coded data sets longer than
// the wave's window of the stream raise kDecRedo like a lane's ring does.  Chosen by launch_decode_any for few, long
// enough RSIs (dec_wave_wanted).
constexpr uint32_t kDwWin = 1024;          // words of the stream a wavefront stages in LDS
constexpr uint32_t kDwPad = 72;            // zero words behind them (register window, peeks of decode_block)
constexpr uint32_t kDwPiece = 4096;        // bits whose coded-data-set lengths are tabulated at a time
// words of LDS per wavefront: window | rank per word of (piece + look-ahead) | positions of the 1-bits of (piece +
// look-ahead) (u16 each)
__host__ __device__ inline uint32_t dw_look_words(const Cfg &c) { return (c.id_len + 1u + c.bs * c.bps + 31u) / 32u + 2u; }
__host__ __device__ inline uint32_t dw_wave_words(const Cfg &c)
{
    const uint32_t pw = kDwPiece / 32u + dw_look_words(c);
    return kDwWin + kDwPad + (pw + 2u + 1u) / 2u + pw * 16u;
}

struct WaveSrc {
    const uint32_t *win;
    uint32_t limit;        // words that hold stream (relative index); reads at or beyond it starve
    uint32_t starve;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return win[i < kDwWin + kDwPad - 1u ? i : kDwWin + kDwPad - 1u]; }
    __device__ __forceinline__ uint32_t word(uint32_t i)
    {
        starve |= (i >= limit) ? 1u : 0u;
        return at(i);
    }
    __device__ __forceinline__ void word2(uint32_t i, uint32_t &w0, uint32_t &w1)
    {
        starve |= (i + 1 >= limit) ? 1u : 0u;
        w0 = at(i);
        w1 = at(i + 1u);
    }
    __device__ __forceinline__ void word3(uint32_t i, uint32_t &w0, uint32_t &w1, uint32_t &w2)
    {
        starve |= (i + 2 >= limit) ? 1u : 0u;
        w0 = at(i);
        w1 = at(i + 1u);
        w2 = at(i + 2u);
    }
    __device__ __forceinline__ bool starved() const { return starve != 0; }
};

struct WaveWinFetch {      // word source of the sequential reader (coded data sets beyond the register window)
    const uint32_t *win;
    uint64_t base;         // stream word index of win[0]
    __device__ __forceinline__ uint32_t operator()(uint64_t idx) const
    {
        const uint64_t rel = idx - base;
        return rel < kDwWin + kDwPad ? win[rel] : 0u;
    }
};

#ifdef AEC_TUNING
__device__ unsigned long long g_dw_prof[8];      // (diagnostics, AEC_DW_PROF=1: shader-clock ticks of RSI 0's phases)
#define DW_T0() const unsigned long long dw_t0 = __builtin_amdgcn_s_memtime()
#define DW_T1(k) do { if (r == 0 && lane == 0) g_dw_prof[k] += __builtin_amdgcn_s_memtime() - dw_t0; } while (0)
#else
#define DW_T0() do { } while (0)
#define DW_T1(k) do { } while (0)
#endif

template <int BS, int BYTES>
__global__ void __launch_bounds__(256)
k_decode_wave(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
              const uint64_t *__restrict__ rsi_off, uint64_t n_rsi, uint64_t total_blocks, uint8_t *__restrict__ out,
              DecResult *res, uint8_t *__restrict__ dump, const DecResult *__restrict__ idx,
              const DecResult *__restrict__ batch, uint32_t rsi_per_chunk)
{
    static_assert(BS == 8 || BS == 16 || BS == 32 || BS == 64, "templated block sizes");
    if (idx) {
        const uint64_t whole = idx->n_rsi, tail = idx->tail_blocks;
        n_rsi = whole + (tail ? 1u : 0u);
        total_blocks = whole * c.rsi + tail;
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *win = smem + (size_t)wave * dw_wave_words(c);
    const uint32_t pwords = kDwPiece / 32u + dw_look_words(c);            // words of a piece + its look-ahead
    uint16_t *prank = reinterpret_cast<uint16_t *>(win + kDwWin + kDwPad);  // [pwords + 1]: 1-bits in front of a word
    uint16_t *ones = reinterpret_cast<uint16_t *>(win + kDwWin + kDwPad + (pwords + 2u + 1u) / 2u);   // [pwords * 32]
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= n_rsi) return;
    const bool pp = c.flags & F_PREPROCESS, sgn = c.flags & F_SIGNED;
    uint32_t nb;
    uint64_t start;
    if (batch) {
        const uint64_t sidx = r / rsi_per_chunk;
        const uint32_t rin = (uint32_t)(r - sidx * rsi_per_chunk);
        const uint64_t whole = batch[sidx].n_rsi, tail = batch[sidx].tail_blocks;
        nb = rin < whole ? c.rsi : (rin == whole ? (uint32_t)tail : 0u);
        start = nb ? rsi_off[r] : 0;
    } else {
        const uint64_t left = total_blocks - r * c.rsi;
        nb = left > c.rsi ? c.rsi : (uint32_t)left;
        start = rsi_off[r];
    }
    const uint64_t first_blk = r * c.rsi;
    constexpr uint32_t BLK = (uint32_t)BS * (uint32_t)BYTES;
    uint8_t *dst = out + (size_t)first_blk * BLK;
    const uint32_t maxbits = c.id_len + 1u + c.bps + c.bs * c.bps, idmax = (1u << c.id_len) - 1u;
    const bool coop = maxbits + 128u <= 2048u;

    // ---- the wave's window of the stream
    uint32_t pc0 = 0xFFFFFFFFu;        // first bit (window-relative) of the tabulated piece; none yet
    uint64_t base = 0;                 // stream word of win[0]
    uint32_t have = 0;                 // words of it that hold stream (the rest: zeros)
    bool to_end = false;               // the window reaches the end of the stream
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto refill = [&](uint64_t from_word) {
        DW_T0();
        base = from_word & ~3ull;
        wave_sync();
        for (uint32_t i = lane * 4u; i < kDwWin; i += 64u * 4u) {
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
        for (uint32_t i = lane; i < kDwPad; i += 64u) win[kDwWin + i] = 0u;
        wave_sync();
        const uint64_t left = nwords > base ? nwords - base : 0u;
        have = left < kDwWin ? (uint32_t)left : kDwWin;
        to_end = left <= kDwWin;
        pc0 = 0xFFFFFFFFu;                 // (the table of a piece goes with the window it was made from)
        DW_T1(0);
    };
    // ---- the register window: lane l holds word wb + l of the LDS window
    uint32_t wb = 0, W = 0;
    bool loaded = false;
    auto load_regs = [&](uint32_t first) {
        wb = first;
        W = win[first + lane < kDwWin + kDwPad ? first + lane : kDwWin + kDwPad - 1u];
        loaded = true;
    };
    auto rdlane = [&](uint32_t v, uint32_t l) {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l));
    };
    auto peek = [&](uint32_t rel) {                                      // 32 bits at register-window bit offset rel
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint64_t two = ((uint64_t)rdlane(W, w) << 32) | rdlane(W, (w + 1u) & 63u);
        return (uint32_t)((two << sh) >> 32);
    };
    auto skip_ones = [&](uint32_t rel, uint32_t n) -> uint32_t {         // offset just behind the n-th 1-bit from rel on
        const uint32_t w = rel >> 5, sh = rel & 31u;
        const uint32_t m = lane < w ? 0u : (lane == w ? W & (0xFFFFFFFFu >> sh) : W);
        const uint32_t pc = (uint32_t)__builtin_popcount(m);
        const uint32_t S = wave_incl_sum_dpp(pc);
        const uint64_t enough = __ballot(S >= n);
        if (enough == 0) return 0xFFFFFFFFu;
        const uint32_t L = (uint32_t)__builtin_ctzll(enough);
        const uint32_t need = n - (rdlane(S, L) - rdlane(pc, L));
        const uint32_t word = rdlane(m, L);
        const uint32_t j = lane & 31u;
        const uint32_t bit = (word >> (31u - j)) & 1u;
        const uint32_t rank = j ? (uint32_t)__builtin_popcount(word >> (32u - j)) : 0u;
        const uint64_t hit = __ballot(lane < 32u && bit && rank + 1u == need);
        return L * 32u + (uint32_t)__builtin_ctzll(hit) + 1u;
    };
    // length of the coded data set at window bit `rel` (0 = not found inside the window), nz = zero-run code or 0
    auto cds_len = [&](uint32_t rel, uint32_t ref, uint32_t &nz) -> uint32_t {
        nz = 0;
        if (coop) {
            if (!loaded || rel < wb * 32u || rel - wb * 32u + maxbits + 64u > 2048u) load_regs(rel >> 5);
            const uint32_t q0 = rel - wb * 32u;
            const uint32_t h = peek(q0);
            const uint32_t id = h >> (32u - c.id_len);
            uint32_t q = q0 + c.id_len;
            if (id == 0u) {
                const uint32_t selb = (h >> (31u - c.id_len)) & 1u;
                q += 1u + ref * c.bps;
                if (selb) {
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
            if (q != 0xFFFFFFFFu) return q - q0;
            nz = 0;
        }
        // large blocks, or a unary part that leaves the register window: the sequential reader over the LDS window
        // (every lane the same walk)
        BitReaderT<WaveWinFetch> br;
        br.init(WaveWinFetch{win, base}, ((uint64_t)base + kDwWin + kDwPad) * 32u, base * 32u + rel);
        const uint32_t il = c.id_len;
        const uint32_t id = br.get(il);
        if (id == 0u) {
            const uint32_t selb = br.get(1);
            if (ref) br.skip(c.bps);
            if (selb) {
                if (!br.skip_unary(c.bs / 2u)) return 0u;
            } else {
                uint32_t fs;
                if (!br.unary(fs)) return 0u;
                nz = fs + 1u;
            }
        } else if (id == idmax) {
            br.skip((uint64_t)c.bs * c.bps);
        } else {
            if (ref) br.skip(c.bps);
            if (!br.skip_unary(c.bs - ref)) return 0u;
            br.skip((uint64_t)(c.bs - ref) * (id - 1u));
        }
        const uint64_t len = br.pos - (base * 32u + rel);
        return len < 0x7FFFFFFFull ? (uint32_t)len : 0u;
    };

    // ---- rank and positions of the 1-bits of a piece: window bits [pc0, pc0 + kDwPiece + look-ahead), pc0 a multiple of 32
    uint32_t tcnt = 0, plim = 0;       // 1-bits of the piece; bits of the window that are stream
    auto build_piece = [&](uint32_t from_bit) {
        DW_T0();
        pc0 = from_bit & ~31u;
        const uint32_t w0 = pc0 >> 5;
        plim = to_end ? (uint32_t)(end_bit - base * 32u < (uint64_t)(kDwWin + kDwPad) * 32u ? end_bit - base * 32u
                                                                                              : (uint64_t)(kDwWin + kDwPad) * 32u)
                      : have * 32u;
        wave_sync();
        uint32_t carry = 0;
        for (uint32_t i0 = 0; i0 < pwords; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const uint32_t wi = w0 + i;
            const uint32_t word = (i < pwords && wi < kDwWin + kDwPad) ? win[wi] : 0u;
            const uint32_t pc = (uint32_t)__builtin_popcount(word);
            const uint32_t incl = wave_incl_sum_dpp(pc);
            if (i < pwords) prank[i + 1u] = (uint16_t)(carry + incl);
            uint32_t at = carry + incl - pc, bits = word;             // positions just behind the 1-bits of this word
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
        wave_sync();
        DW_T1(1);
    };
    // the coded data set that would begin at window bit q (pc0 <= q < pc0 + kDwPiece), WITHOUT a reference sample: its
    // entry in the nxt[] format of aec_spec.h -- length | kind, 0 = not resolved here.  One code path for all options:
    // the end of the unary part is the n-th 1-bit behind the header, ones[rank(q1) + n - 1].
    auto fentry = [&](uint32_t q) -> uint32_t {
        const uint32_t il = c.id_len, w = q >> 5, sh = q & 31u;
        const uint32_t wa = w < kDwWin + kDwPad - 1u ? w : kDwWin + kDwPad - 2u;
        const uint32_t a = win[wa], bw = win[wa + 1u];
        const uint32_t h = (uint32_t)(((((uint64_t)a) << 32) | bw) << sh >> 32);
        const uint32_t id = h >> (32u - il);
        const bool unc = id == idmax, low = id == 0u;
        const uint32_t selb = (h >> (31u - il)) & 1u;
        const uint32_t off1 = il + (low ? 1u : 0u);
        const uint32_t n = low ? (selb ? c.bs / 2u : 1u) : c.bs;
        const uint32_t rq = (uint32_t)prank[w - (pc0 >> 5)] + (sh ? (uint32_t)__builtin_popcount(a >> (32u - sh)) : 0u);
        const uint32_t k = rq + (uint32_t)__builtin_popcount(h >> (32u - off1)) + n - 1u;
        const uint32_t e = k < tcnt ? (uint32_t)ones[k] : 0u;
        const uint32_t add = low ? 0u : c.bs * (id - 1u);
        const uint32_t end = unc ? q + il + c.bs * c.bps : e + add;
        const bool ok = (unc || e != 0u) && end <= plim && end - q < 4096u;
        return ok ? ((end - q) | ((low && !selb) ? kNxtZero : kNxtBlock)) : 0u;
    };

    uint64_t pos = start;              // absolute bit of the next coded data set
    uint32_t b = 0;                    // blocks of the RSI located so far
    uint32_t zpend = 0;                // blocks of a zero run that the round in front could not hold
    uint32_t xcarry = 0;               // predictor state behind the blocks stored so far
    uint32_t done_blocks = 0;
    bool stop = false;
    refill(pos >> 5);
    while (done_blocks < nb && !stop) {
        // ---- walk: up to 64 blocks
        DW_T0();
        uint32_t cnt = 0, mypos = 0, myblk = 0;
        bool myparse = false;
        uint32_t fail_lane = 64u;      // the lane whose coded data set the walk could not take: its decode says why
        if (zpend) {                   // the rest of a zero run: blocks of zeros, nothing to parse
            const uint32_t take = zpend < 64u ? zpend : 64u;
            if (lane < take) myblk = done_blocks + lane;
            cnt = take;
            zpend -= take;
        }
        while (cnt < 64u && b < nb && !zpend) {
            uint32_t rel = (uint32_t)(pos - base * 32u);
            // The common steps in a loop of their own: the 64 lanes parse the coded data sets that WOULD begin at the
            // next 64 bits (one each, no dependence between them), then the chain through those 64 bits is a lane read per
            // coded data set -- ~3 of them per round at 23 bits each.  Taken while the coded data sets have one block and
            // an entry; everything else below (reference sample, zero runs, the next piece, the next window, failures).
            if (pc0 != 0xFFFFFFFFu && b != 0u) {
                uint32_t fr = rel, fc = cnt, fb = b;
                bool plain = true;
                while (plain && fr >= pc0 && fr + 64u <= pc0 + kDwPiece) {
                    const uint32_t E = fentry(fr + lane);
                    uint32_t l = 0;
                    do {
                        const uint32_t e = rdlane(E, l);
                        if ((e >> 12) != (kNxtBlock >> 12)) {
                            plain = false;
                            break;
                        }
                        const bool mine = lane == fc;
                        mypos = mine ? fr + l : mypos;
                        myparse = myparse || mine;
                        l += e & 0xFFFu;
                        fc++;
                        fb++;
                    } while (l < 64u && fc < 64u && fb < nb);
                    fr += l;
                    if (fc == 64u || fb == nb) break;
                }
                pos += fr - rel;
                cnt = fc;
                b = fb;
                rel = fr;
                if (cnt == 64u || b == nb) break;
            }
            // (the whole coded data set inside the window -- or the window reaches the end of the stream)
            if (!to_end && rel + maxbits + 64u > have * 32u) {
                if (cnt) break;                                   // decode what the round holds, refill behind it
                refill(pos >> 5);
                loaded = false;
                rel = (uint32_t)(pos - base * 32u);
            }
            const uint32_t ref = (pp && b == 0u) ? 1u : 0u;
            uint32_t nz = 0, len = 0;
            if (!ref) {
                // (the piece must have its look-ahead inside the window's stream, or the window reach the stream's end)
                if (pc0 == 0xFFFFFFFFu || rel < pc0 || rel >= pc0 + kDwPiece) {
                    const uint32_t from = rel & ~31u;
                    if (!to_end && from + kDwPiece + dw_look_words(c) * 32u > have * 32u && from > 4096u) {
                        if (cnt) break;                           // (a fresh window for a fresh piece)
                        refill(pos >> 5);
                        loaded = false;
                        rel = (uint32_t)(pos - base * 32u);
                    }
                    build_piece(rel);
                }
                const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)fentry(rel));
                len = e & 0xFFFu;
                nz = (e & kNxtZero) ? len - c.id_len - 1u : 0u;
            }
            if (!len) len = cds_len(rel, ref, nz);                // (reference sample, or beyond the table's look-ahead)
            const uint32_t nblk = len ? (nz ? spec_run_blocks(c, nz, b) : 1u) : 0u;
            const bool inside = len != 0u && (to_end ? base * 32u + rel + len <= end_bit : rel + len <= have * 32u);
            if (!inside || !nblk) {
                // A coded data set that does not end inside a freshly filled window is longer than any an encoder of
                // the reference writes -- or the stream is cut or damaged here.  The lane's decode gives the verdict
                // (status, or kDecRedo when it ran out of the window) exactly as a lane of k_decode would.
                if (!to_end && cnt) break;
                if (lane == cnt) {
                    mypos = rel;
                    myblk = b;
                    myparse = true;
                }
                fail_lane = cnt;
                cnt++;
                stop = true;
                break;
            }
            uint32_t take = nblk;
            if (take > nb - b) take = nb - b;                     // (a caller's block count may cut a run)
            const uint32_t room = 64u - cnt;
            const uint32_t now = take < room ? take : room;
            if (lane >= cnt && lane < cnt + now) {
                mypos = rel;
                myblk = b + (lane - cnt);
                myparse = lane == cnt;
            }
            zpend = take - now;
            cnt += now;
            pos += len;
            b += take;
        }
        DW_T1(2);                      // (includes the refills and pieces inside the walk)
        myblk = done_blocks + lane;    // (the blocks of a round are consecutive)
        // ---- decode: lane i, block done_blocks + i
        uint32_t d[BS];
        const bool live = lane < cnt;
        const uint32_t ref_l = (pp && live && myparse && myblk == 0u) ? 1u : 0u;
        WaveSrc src{win, to_end ? 0xFFFFFFFFu : have, 0u};
        uint32_t p = mypos, nzl = 0;
        const uint64_t left_bits = end_bit - base * 32u;
        const uint32_t end_p = left_bits > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)left_bits;
        uint32_t st = decode_block<BS>(src, p, end_p, d, c, ref_l, myblk, live && myparse, nzl);
        if (!(live && myparse)) {
#pragma unroll
            for (int i = 0; i < BS; i++) d[i] = 0u;
            st = DEC_OK;
        }
        const bool over = live && myparse && !to_end && (((p + 31u) >> 5) > have || (st != DEC_OK && src.starved()));
        DW_T1(3);                      // (walk + decode)
        const uint64_t bad = __ballot((live && myparse && st != DEC_OK) || over);
        uint32_t good = cnt;           // blocks of this round that are delivered
        if (bad) {
            const uint32_t f = (uint32_t)__builtin_ctzll(bad);
            good = f;
            stop = true;
            if (lane == f) {
                if (over) {
                    atomicOr(&res->pad, kDecRedo);
                } else {
                    report(res, st, r, first_blk + done_blocks + f);
                    if (batch && st == DEC_DATA_ERROR)
                        atomicMax(&const_cast<DecResult *>(batch)[r / rsi_per_chunk].status, (uint32_t)DEC_DATA_ERROR);
                }
            }
        } else if (fail_lane < 64u) {
            // (the walk could not take a coded data set that the decode accepts: cannot happen -- both read the same
            // bits -- but must not go unnoticed: the batch is decoded again by the sequential path)
            if (lane == 0) atomicOr(&res->pad, kDecRedo);
            good = fail_lane;
            stop = true;
        }
        // ---- inverse predictor + store
        const bool act = lane < good;
        uint8_t *q = act ? dst + (size_t)(done_blocks + lane) * BLK : dump + (size_t)lane * BLK;
        uint32_t x = 0;
        if (!pp) {
            store_block<BS, BYTES>(q, d, c, false, x);
        } else {
            // the running sum of the steps (reference decode.c:96-134: +d/2 or -(d+1)/2 while nothing clips)
            uint32_t ssum = 0;
#pragma unroll
            for (int i = 0; i < BS; i++) {
                const uint32_t v = (i == 0 && ref_l) ? 0u : d[i];
                ssum += (v >> 1) ^ (0u - (v & 1u));
            }
            if (!act) ssum = 0;
            const uint32_t incl = wave_incl_sum_dpp(ssum);
            const uint32_t has_ref = (uint32_t)__builtin_amdgcn_readlane((int)ref_l, 0);
            const uint32_t refv = (uint32_t)__builtin_amdgcn_readlane((int)d[0], 0);
            const uint32_t xb = has_ref ? (sgn ? sign_extend(refv, c.bps) : refv) : xcarry;
            uint32_t xin = xb + incl - ssum;
            uint32_t xout;
            for (;;) {
                xout = xin;
                store_block<BS, BYTES>(q, d, c, ref_l != 0u, xout);
                const uint32_t prev = (uint32_t)__shfl_up((int)xout, 1);        // lane l - 1's output (lane 0: its own)
                const uint64_t wrong = __ballot(act && lane != 0u && !ref_l && prev != xin);
                if (!wrong) break;
                const uint32_t f = (uint32_t)__builtin_ctzll(wrong);
                const uint32_t delta = rdlane(prev - xin, f);
                if (lane >= f) xin += delta;
            }
            if (good) xcarry = rdlane(xout, good - 1u);
            x = xcarry;
        }
        done_blocks += good;
        DW_T1(4);                      // (walk + decode + predictor + store)
        if (bad || fail_lane < 64u) break;
    }
    // the predictor state behind the LAST item of the batch (k_decode_partial continues from it)
    if (lane == 0 && r + 1 == n_rsi) res->end_bit = pp ? xcarry : 0u;
}

#ifndef AEC_DEC_PART                  // (what does not depend on the block size: the one object without a part)
// ---- the coded data set the input ends in ------------------------------------------------------------
// The reference's resumable readers release every sample whose bits have arrived, also from a coded
// data set that is cut by the end of the input (reference src/decode.c:342-400 bits_ask / fs_ask,
// :423-460 m_split_output / m_split_fs, :560-587 m_se_decode, :646-657 m_uncomp_copy): the
// reference sample as soon as its bits are there, second-extension pairs code by code, uncompressed
// samples one by one, split samples once ALL fundamental sequences of the block are in, one per
// k-bit field.  The block-parallel kernel above only handles complete coded data sets; this
// single-lane kernel adds the samples of the ONE incomplete coded data set behind them (the index
// pass says where it starts: idx->end_bit, block idx->tail_blocks of RSI idx->n_rsi of the batch).
__global__ void __launch_bounds__(64)
k_decode_partial(const Cfg c, const uint32_t *__restrict__ words, uint64_t nwords, uint64_t end_bit,
                 const DecResult *__restrict__ idx, uint8_t *__restrict__ out, DecResult *res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (idx->pad != 1u || idx->status != DEC_OK) return;     // the walk did not stop for lack of input
    const uint32_t bs = c.bs, bps = c.bps;
    const bool pp = c.flags & F_PREPROCESS, sgn = c.flags & F_SIGNED, msb = c.flags & F_MSB;
    const uint32_t b = (uint32_t)idx->tail_blocks;
    const uint32_t ref = (pp && b == 0) ? 1u : 0u;
    const uint32_t redo_flag = res->pad & kDecRedo;          // (k_decode's note that the slow path ran: kept)
    BitReader r;
    r.init(words, nwords, end_bit, idx->end_bit);
    auto has = [&](uint32_t n) { return r.pos + n <= end_bit; };
    auto unary_ok = [&](uint32_t &z) { return r.unary(z) && r.pos <= end_bit; };
    uint32_t d[kMaxBlockSize];
    uint32_t cnt = 0;
    do {
        if (!has(c.id_len)) break;
        const uint32_t id = r.get(c.id_len);
        if (id == 0) {
            if (!has(1)) break;
            const uint32_t sel = r.get(1);
            if (ref) {
                if (!has(bps)) break;
                d[cnt++] = r.get(bps);
            }
            if (!sel) break;                                 // a zero-block code is all or nothing
            uint32_t i = ref;
            while (i < bs) {
                uint32_t m = 0, sum = 0, second = 0;
                if (!unary_ok(m)) break;
                if (!se_lookup(m, sum, second)) {            // beyond the table: corrupt (decode.c:589-616)
                    report(res, DEC_DATA_ERROR, idx->n_rsi, idx->n_rsi * c.rsi + idx->tail_blocks);
                    break;
                }
                if ((i & 1u) == 0) d[cnt++] = sum - second, i++;
                d[cnt++] = second;
                i++;
            }
        } else if (id == (1u << c.id_len) - 1u) {
            while (cnt < bs && has(bps)) d[cnt++] = r.get(bps);
        } else {
            const uint32_t k = id - 1u;
            if (ref) {
                if (!has(bps)) break;
                d[cnt++] = r.get(bps);
            }
            uint32_t fs[kMaxBlockSize];
            const uint32_t n = bs - ref;
            uint32_t got = 0;
            while (got < n && unary_ok(fs[got])) got++;
            if (got < n) break;
            for (uint32_t i = 0; i < n && has(k); i++) d[cnt++] = (fs[i] << k) + (k ? r.get(k) : 0u);
        }
    } while (false);
    if (cnt >= bs) cnt = bs - 1;                             // (a complete block is the index pass's business)
    // inverse predictor (reference decode.c:67-141) continuing from the sample in front, byte-order store
    const uint64_t first = (idx->n_rsi * c.rsi + b) * (uint64_t)bs;       // sample index in `out`
    // (the predictor state k_decode left behind the blocks in front -- not the sample in the output, which is cut to
    // the sample width)
    uint32_t x = 0;
    if (pp && !ref && cnt) x = (uint32_t)res->end_bit;
    for (uint32_t i = 0; i < cnt; i++) {
        uint32_t v;
        if (!pp) v = d[i];
        else if (i == 0 && ref) v = x = sgn ? sign_extend(d[0], bps) : d[0];
        else v = x = sgn ? unpp_signed(x, d[i], c.xmax) : unpp_unsigned(x, d[i], c.xmax);
        for (uint32_t t = 0; t < c.bytes; t++)
            out[(first + i) * c.bytes + t] = (uint8_t)(v >> (8 * (msb ? c.bytes - 1 - t : t)));
    }
    res->pad = cnt | redo_flag;
}

__global__ void k_dec_result_init(DecResult *res)
{
    res->n_rsi = 0;
    res->tail_blocks = ~0ull;          // (decode records: the lowest failing block of the batch, see report())
    res->end_bit = 0;
    res->status = DEC_OK;
    res->pad = 0;
    res->bad_rsi = ~0ull;
}

// 64 lanes x one maximal block: where lanes without a block to write send their (always issued)
// stores.  One buffer per device for the life of the process.
uint8_t *dump_buffer()
{
    static std::mutex mu;
    static uint8_t *buf[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!buf[dev] && hipMalloc(reinterpret_cast<void **>(&buf[dev]), 64 * kMaxBlockSize * 4) != hipSuccess) {
        buf[dev] = nullptr;            // the caller reports the failure (idle lanes would store through it)
        (void)hipGetLastError();
    }
    return buf[dev];
}
#endif

struct DecGeom {
    uint32_t ring_words, maxw, needw, waves, grid;
    size_t lds_bytes;
};

// Ring capacity: a coded data set never exceeds id_len + 1 + bps + bs*bps bits (maxw words with the
// look-ahead of the 64-bit peeks); with loads landing one iteration late a ring of two of them plus
// alignment slack never starves.  LDS is what bounds the waves per SIMD of k_decode, so when the
// stream averages short coded data sets (avg_cds_bits, from the caller's byte and block counts) and
// one maximal CDS still fits, the ring is HALF that: steady state keeps needw words ahead, a longer
// CDS is decoded again after a full refill (k_decode).  Incompressible input keeps the full ring.
DecGeom dec_geom(const Cfg &c, uint64_t n_rsi, uint64_t avg_cds_bits, uint32_t stg_row_bytes)
{
    DecGeom g;
    // (blocks of 8 samples are decoded AEC_DEC_UNR8 per top-up of the ring)
    const uint32_t maxbits = (c.id_len + 1 + c.bps + c.bs * c.bps) * dec_unroll((int)c.bs);
    g.maxw = maxbits / 32 + 5;   // + look-ahead of the 64-bit peeks
    uint32_t rw = 16;
    while (rw < 2 * g.maxw + 3) rw <<= 1;
    g.needw = g.maxw;
    // half ring: after a full refill at least rw/2 - 3 words lie ahead of any position
    // (how many average coded data sets half of the half ring -- its steady-state look-ahead -- must hold: 2.  With
    // 4, typical.dat's 720-bit blocks of 64 samples kept the full ring, 33 KB per wave and 4 waves per CU: 5.8 ms;
    // on the half ring 3.3 ms, the second attempts included.  AEC_DEC_HALF_FACTOR overrides, for measurements.)
    const uint64_t hf = tune("AEC_DEC_HALF_FACTOR", 2);
    if (rw >= 32 && g.maxw <= rw / 2 - 3 && avg_cds_bits && avg_cds_bits * hf <= (uint64_t)rw / 2 * 32) {
        rw /= 2;
        g.needw = rw / 2;
    }
    g.ring_words = rw;
    // (+ the output staging rows of k_decode for small blocks: 64 x (row + 16) bytes)
    // (AEC_DEC_LDS_PAD: extra LDS bytes per wave, a diagnostic knob for occupancy experiments; the kernel
    // never touches them)
    const size_t pad = tune("AEC_DEC_LDS_PAD", 0);
    const size_t per_wave = (size_t)(rw + 2) * 64 * 4 + (stg_row_bytes ? 64 * (stg_row_bytes + 16) : 0) + pad;
    // waves per workgroup: whatever packs most waves into the 160 KiB of a CU
    uint32_t waves = 1, best = 0;
    for (uint32_t w = 1; w <= 4; w *= 2) {
        if (per_wave * w > 65536) break;
        uint32_t fit = (uint32_t)(160 * 1024 / (per_wave * w)) * w;
        if (fit > 32) fit = 32;                         // a CU holds 32 waves at most
        if (fit >= best) {                              // ties: the larger workgroup
            best = fit;
            waves = w;
        }
    }
    g.waves = waves;
    g.lds_bytes = per_wave * waves;
    const uint64_t nwaves = (n_rsi + 63) / 64;
    g.grid = (uint32_t)((nwaves + waves - 1) / waves);
    return g;
}

template <int BS, bool SEG, bool SUMS = false>
void launch_decode_bytes(const Cfg &c, const uint32_t *words, uint64_t nwords, uint64_t end_bit,
                         const uint64_t *rsi_off, const SegEntry *seg_table, uint64_t n_rsi,
                         uint64_t total_blocks, uint8_t *out, DecResult *res, hipStream_t st, uint8_t *dump,
                         const DecResult *idx, const DecResult *batch, uint32_t rpc, const BareArgs &ba = BareArgs())
{
    const uint32_t blk = (uint32_t)BS * c.bytes;
    // (counts taken from the index record: the average coded data set is not known here unless the caller says
    // so -- full ring)
    const uint64_t avg = (total_blocks && !idx && !batch) ? end_bit / total_blocks : ba.avg_hint;
    const DecGeom g = dec_geom(c, n_rsi, avg, (!SUMS && stg_on((int)blk)) ? stg_row((int)blk) : 0u);
    const dim3 block(64 * g.waves), grid(g.grid);
    // loads in flight per block iteration (see kPend): sized for the average coded data set where the caller
    // knows it, for the worst case of large blocks where it does not
    const uint32_t e_kp = tune("AEC_DEC_KP", 0);               // (diagnostic: force 2, 4 or 8)
    // (measured: 2 up to the 256 bits per block they feed -- C3 at 247: 2.70 ms against 2.74 with 4 --, 8 for the
    // 720 bits of typical.dat's blocks: 5.8 ms against 7.1 with 4 and 7.8 with 2)
    int kp = BS >= 32 ? (avg == 0 || avg > 512 ? 8 : (avg > 256 ? 4 : 2)) : (avg > 256 ? 4 : 2);
    if (e_kp) kp = e_kp >= 8 ? (BS >= 32 ? 8 : 4) : (e_kp >= 4 ? 4 : 2);
#define AEC_GO2(B, KP)                                                                                  \
    hipLaunchKernelGGL((k_decode<BS, B, SEG, KP, SUMS>), grid, block, g.lds_bytes, st, c, words, nwords, end_bit, \
                       rsi_off, seg_table, n_rsi, total_blocks, out, res, g.ring_words, g.maxw, g.needw, dump, idx, batch, rpc, \
                       ba.sums, ba.list, ba.list_cnt)
#define AEC_GO(B)                                                                                   \
    do {                                                                                            \
        if (kp == 2) AEC_GO2(B, 2);                                                                 \
        else if (kp == 4) AEC_GO2(B, 4);                                                            \
        else if (BS >= 32) AEC_GO2(B, (BS >= 32 ? 8 : 4));                                          \
    } while (0)
    switch (c.bytes) {
    case 1: AEC_GO(1); break;
    case 2: AEC_GO(2); break;
    case 3: AEC_GO(3); break;
    default: AEC_GO(4); break;
    }
#undef AEC_GO
#undef AEC_GO2
}

}  // namespace

#ifdef AEC_DEC_PART
// ---- the kernels of ONE block size (this object: -DAEC_DEC_PART=<block size>, 0 = any other) --------------------------------
template <int BS>
void dec_part_bytes(bool seg, bool sums, const DecLaunch &a)
{
    const Cfg &c = *a.c;
    if constexpr (BS == 0) {
        // generic block sizes and containers: the sample-by-sample reader has no second attempt: full ring
        const DecGeom g = dec_geom(c, a.n_items, 0, 0u);
        if (seg)
            hipLaunchKernelGGL((k_decode<0, 0, true, 2>), dim3(g.grid), dim3(64 * g.waves), g.lds_bytes, a.st, c, a.words, a.nwords,
                               a.end_bit, a.rsi_off, a.seg_table, a.n_items, a.total_blocks, a.out, a.res, g.ring_words, g.maxw,
                               g.needw, a.dump, a.idx, a.batch, a.rpc, (SegSum *)nullptr, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_decode<0, 0, false, 2>), dim3(g.grid), dim3(64 * g.waves), g.lds_bytes, a.st, c, a.words, a.nwords,
                               a.end_bit, a.rsi_off, a.seg_table, a.n_items, a.total_blocks, a.out, a.res, g.ring_words, g.maxw,
                               g.needw, a.dump, a.idx, a.batch, a.rpc, (SegSum *)nullptr, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr);
    } else if (sums) {
        launch_decode_bytes<BS, true, true>(c, a.words, a.nwords, a.end_bit, a.rsi_off, a.seg_table, a.n_items, a.total_blocks,
                                            a.out, a.res, a.st, a.dump, a.idx, a.batch, a.rpc, a.ba);
    } else if (seg) {
        launch_decode_bytes<BS, true, false>(c, a.words, a.nwords, a.end_bit, a.rsi_off, a.seg_table, a.n_items, a.total_blocks,
                                             a.out, a.res, a.st, a.dump, a.idx, a.batch, a.rpc, a.ba);
    } else {
        launch_decode_bytes<BS, false, false>(c, a.words, a.nwords, a.end_bit, a.rsi_off, a.seg_table, a.n_items, a.total_blocks,
                                              a.out, a.res, a.st, a.dump, a.idx, a.batch, a.rpc, a.ba);
    }
}

template <int BS>
void dec_part_wave(const DecLaunch &a)
{
    const Cfg &c = *a.c;
    constexpr int B = BS;
    const uint32_t waves = 2;                            // (~24 KB of LDS per wavefront: six of them on a CU)
    const dim3 grid((uint32_t)((a.n_items + waves - 1) / waves)), block(64 * waves);
    const size_t lds = (size_t)waves * dw_wave_words(c) * 4;
#define AEC_WV2(BY)                                                                                                     \
    hipLaunchKernelGGL((k_decode_wave<B, BY>), grid, block, lds, a.st, c, a.words, a.nwords, a.end_bit, a.rsi_off, a.n_items, \
                       a.total_blocks, a.out, a.res, a.dump, a.idx, a.batch, a.rpc)
#ifdef AEC_TUNING
    static const bool dw_prof = tune("AEC_DW_PROF", 0) != 0;
    if (dw_prof) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_dw_prof), z, sizeof(z), 0, hipMemcpyHostToDevice, a.st);
    }
#endif
    switch (c.bytes) {
    case 1: AEC_WV2(1); break;
    case 2: AEC_WV2(2); break;
    case 3: AEC_WV2(3); break;
    default: AEC_WV2(4); break;
    }
#undef AEC_WV2
#ifdef AEC_TUNING
    if (dw_prof) {
        static int reports = 0;
        if (reports++ < 6) {
            unsigned long long h[8];
            (void)hipStreamSynchronize(a.st);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dw_prof), sizeof(h));
            fprintf(stderr, "k_decode_wave, RSI 0 (s_memtime ticks): refills %llu | pieces %llu | walk incl. those %llu | "
                    "+ decode %llu | + predictor, store %llu\n", h[0], h[1], h[2], h[3], h[4]);
        }
    }
#endif
}

template void dec_part_bytes<AEC_DEC_PART>(bool, bool, const DecLaunch &);
#if AEC_DEC_PART != 0
template void dec_part_wave<AEC_DEC_PART>(const DecLaunch &);
#endif

#else       // ---- the object without a part: dispatch, and what does not depend on the block size ---------------------
extern template void dec_part_bytes<0>(bool, bool, const DecLaunch &);
extern template void dec_part_bytes<8>(bool, bool, const DecLaunch &);
extern template void dec_part_bytes<16>(bool, bool, const DecLaunch &);
extern template void dec_part_bytes<32>(bool, bool, const DecLaunch &);
extern template void dec_part_bytes<64>(bool, bool, const DecLaunch &);
extern template void dec_part_wave<8>(const DecLaunch &);
extern template void dec_part_wave<16>(const DecLaunch &);
extern template void dec_part_wave<32>(const DecLaunch &);
extern template void dec_part_wave<64>(const DecLaunch &);

static void dec_part_bytes_bs(uint32_t bs, bool seg, bool sums, const DecLaunch &a)
{
    switch (bs) {
    case 8: dec_part_bytes<8>(seg, sums, a); break;
    case 16: dec_part_bytes<16>(seg, sums, a); break;
    case 32: dec_part_bytes<32>(seg, sums, a); break;
    case 64: dec_part_bytes<64>(seg, sums, a); break;
    default: dec_part_bytes<0>(seg, sums, a); break;
    }
}

// Few RSIs, each long enough to keep a wavefront's lanes busy: a wavefront per RSI (k_decode_wave) instead of a lane.
// A lane per RSI needs ~260 000 RSIs to fill the chip and takes as long as ONE RSI's serial chain however few there
// are; a wavefront per RSI costs ~7 times the instructions per RSI, which only matters once the chip is full.
static bool dec_wave_wanted(const Cfg &c, uint64_t n_items)
{
    if (c.bs != 8u && c.bs != 16u && c.bs != 32u && c.bs != 64u) return false;
    // (a wavefront takes ~40 us for an RSI of 128 blocks and ~1500 of them run at a time: from ~8000 RSIs on the lane
    // per RSI, 205 us whatever the number, is through first -- 16 MiB of the 8-bit shape: 1.53 against 1.57 ms per call.
    // n_items is the MOST the index pass can have found where the counts come from its record: the streaming ABI asks
    // for room / RSI size + 2 of at least 4 MiB of room, 4098 for the 8-bit shape whatever the call holds)
    // (RSIs of 64 .. 127 blocks: as many more as they are shorter -- the bound of a small call is 8194 for RSIs of 64
    // blocks of 8 samples, which left a 64 KiB chunk of 128 RSIs to a lane per RSI: 102 us of the call's 260)
    const uint32_t most = tune("AEC_DEC_WAVE_MAX", 8192u), least = tune("AEC_DEC_WAVE_RSI", 16u);
    const uint64_t scaled = c.rsi >= 64u && c.rsi < 128u ? (uint64_t)most * 128u / c.rsi : most;
    return c.rsi >= least && n_items <= scaled;
}

template <bool SEG>
static bool launch_decode_any(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_rsi_off,
                              const SegEntry *d_seg, uint64_t n_items, uint64_t total_blocks, uint8_t *d_out,
                              DecResult *d_res, hipStream_t st, const PhaseEvents *prof,
                              const DecResult *d_idx = nullptr, const DecResult *d_batch = nullptr, uint32_t rpc = 0)
{
    uint8_t *dump = dump_buffer();
    if (!dump) return false;
    hipLaunchKernelGGL(k_dec_result_init, dim3(1), dim3(1), 0, st, d_res);
    if (n_items == 0) return true;
    if (prof) (void)hipEventRecord(prof->ev[5], st);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(d_in);
    const uint64_t nwords = (in_bytes + 3) / 4;
    const uint64_t end_bit = (uint64_t)in_bytes * 8;
    // vector stores need 16-byte aligned blocks
    const bool vec_ok = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
    const uint32_t bs = vec_ok ? c.bs : 0;
    DecLaunch a{&c, words, nwords, end_bit, d_rsi_off, d_seg, n_items, total_blocks, d_out, d_res, st, dump, d_idx, d_batch, rpc,
                BareArgs()};
    if (!SEG && bs && dec_wave_wanted(c, n_items)) {
        switch (bs) {
        case 8: dec_part_wave<8>(a); break;
        case 16: dec_part_wave<16>(a); break;
        case 32: dec_part_wave<32>(a); break;
        default: dec_part_wave<64>(a); break;
        }
    } else {
        dec_part_bytes_bs(bs, SEG, false, a);
    }
    if (prof) (void)hipEventRecord(prof->ev[6], st);
    // (behind the timed kernel: returns at once unless k_decode raised kDecRedo)
    const uint64_t redo_waves = (n_items + 63) / 64;
    hipLaunchKernelGGL((k_decode_redo<SEG>), dim3((uint32_t)(redo_waves < 2048 ? redo_waves : 2048)), dim3(64), 0, st, c,
                       words, nwords, end_bit, d_rsi_off, d_seg, n_items, total_blocks, d_out, d_res, d_idx, d_batch, rpc,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    return true;
}

// ---- a bare stream segment by segment ------------------------------------------------------------------
// One lane per RSI: are its segments all known, parsed, and does each end where the next one starts (the last
// one where the next RSI does)?  Does the running sum hold for every one of them, given the sample the sums of
// the segments in front lead to?  Then its entries of the segment table are filled in -- start bit and the sample
// in front, what k_decode<SEG> takes.  If not (a zero run across a segment border, a sample at the edge of the
// range where the predictor clips, decode.c:96-134, a coded data set beyond the look-ahead, anything damaged) the
// RSI goes to the list of those that are decoded by one lane from their start, and its entries say so (~0).
__global__ void __launch_bounds__(256)
k_seg_scan(const Cfg c, const uint64_t *__restrict__ rsi_off, const uint64_t *__restrict__ seg_bits,
           const SegSum *__restrict__ sums, uint64_t n_rsi, uint64_t total_blocks, const DecResult *__restrict__ idx,
           SegEntry *__restrict__ table, uint32_t *__restrict__ list, uint32_t *__restrict__ list_cnt)
{
    if (idx) {
        const uint64_t whole = idx->n_rsi, tail = idx->tail_blocks;
        n_rsi = whole + (tail ? 1u : 0u);
        total_blocks = whole * c.rsi + tail;
    }
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rsi) return;
    uint64_t left = total_blocks > r * c.rsi ? total_blocks - r * c.rsi : 0u;
    if (left > c.rsi) left = c.rsi;
    const uint32_t nseg = (uint32_t)((left + 63u) / 64u), S = c.segs_per_rsi;
    const uint64_t *bits = seg_bits + r * S;
    const SegSum *su = sums + r * S;
    SegEntry *e = table + r * S;
    bool good = nseg != 0u && bits[0] == rsi_off[r];
    for (uint32_t j = 0; j < nseg && good; j++) {
        good = bits[j] != ~0ull && su[j].ok != 0u;
        if (good && j + 1u < nseg) good = su[j].end == bits[j + 1u];
    }
    if (good) {
        const uint64_t end = su[nseg - 1u].end;
        if (r + 1u < n_rsi) good = ((c.flags & F_PAD_RSI) ? (end + 7u) & ~7ull : end) == rsi_off[r + 1u];
        else if (idx) good = end == idx->end_bit;
    }
    const uint32_t mask = c.bps >= 32u ? 0xFFFFFFFFu : (1u << c.bps) - 1u;
    if (good) {
        const bool sgn = c.flags & F_SIGNED;
        const uint32_t ref = su[0].ref;
        // u = sample - xmin (xmin = -2^(bps-1) for signed samples, 0 else)
        int64_t u = sgn ? (int64_t)(int32_t)sign_extend(ref, c.bps) + ((int64_t)1 << (c.bps - 1u)) : (int64_t)ref;
        const int64_t xmin = sgn ? -((int64_t)1 << (c.bps - 1u)) : 0;
        for (uint32_t j = 0; j < nseg; j++) {
            if ((c.flags & F_PREPROCESS) && (u < su[j].lo || u > su[j].hi)) {
                good = false;
                break;
            }
            e[j].bit = bits[j];
            e[j].prev = j ? (uint32_t)(uint64_t)(u + xmin) & mask : 0u;
            e[j].pad = 0;
            u += su[j].sum;
        }
    }
    if (!good) {
        for (uint32_t j = 0; j < S; j++) e[j] = SegEntry{~0ull, 0u, 0u};
        list[atomicAdd(list_cnt, 1u)] = (uint32_t)r;
    } else {
        for (uint32_t j = nseg; j < S; j++) e[j] = SegEntry{~0ull, 0u, 0u};
    }
}

bool launch_decode(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_rsi_off,
                   uint64_t n_rsi, uint64_t total_blocks, uint8_t *d_out, DecResult *d_res, hipStream_t st,
                   const PhaseEvents *prof, const DecResult *d_idx, const DecResult *d_batch, uint32_t rsi_per_chunk)
{
    return launch_decode_any<false>(c, d_in, in_bytes, d_rsi_off, nullptr, n_rsi, total_blocks, d_out, d_res, st,
                                    prof, d_idx, d_batch, rsi_per_chunk);
}

void launch_decode_partial(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const DecResult *d_idx, uint8_t *d_out,
                           DecResult *d_res, hipStream_t st)
{
    hipLaunchKernelGGL(k_decode_partial, dim3(1), dim3(64), 0, st, c, reinterpret_cast<const uint32_t *>(d_in),
                       (uint64_t)((in_bytes + 3) / 4), (uint64_t)in_bytes * 8, d_idx, d_out, d_res);
}

// Workspace of launch_decode_bare for up to max_rsi RSIs: sums and table per segment, list per RSI, counter.
bool decode_bare_supported(const Cfg &c)
{
    // (from eight segments per RSI on: an RSI in which a sample comes within reach of the ends of the range falls
    // back to one lane, and with the reference's sample file -- four segments per RSI -- that is every other RSI;
    // the templated block sizes)
    return c.segs_per_rsi >= 8u && (c.bs == 8u || c.bs == 16u || c.bs == 32u || c.bs == 64u);
}

size_t decode_bare_workspace_bytes(const Cfg &c, uint64_t max_rsi)
{
    const uint64_t nseg = max_rsi * c.segs_per_rsi;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(nseg * sizeof(SegSum)) + up(nseg * sizeof(SegEntry)) + up(max_rsi * 4) + 256;
}

// A bare stream whose index pass also found the segment starts (d_seg_bits[r * segs_per_rsi + j], ~0 = unknown):
// summing pass, scan per RSI, one lane per segment for the RSIs that can be taken that way, one lane per RSI for
// the rest.  Same contract as launch_decode otherwise (d_idx: counts from the index record on the device).
bool launch_decode_bare(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_rsi_off,
                        const uint64_t *d_seg_bits, uint64_t n_rsi, uint64_t total_blocks, uint8_t *d_out,
                        DecResult *d_res, hipStream_t st, const PhaseEvents *prof, const DecResult *d_idx, void *d_ws,
                        size_t ws_bytes, uint64_t avg_cds_hint)
{
    const bool vec_ok = (reinterpret_cast<uintptr_t>(d_out) & 15u) == 0;
    if (!decode_bare_supported(c) || !vec_ok || !d_seg_bits || !d_ws || ws_bytes < decode_bare_workspace_bytes(c, n_rsi))
        return launch_decode(c, d_in, in_bytes, d_rsi_off, n_rsi, total_blocks, d_out, d_res, st, prof, d_idx);
    uint8_t *dump = dump_buffer();
    if (!dump) return false;
    hipLaunchKernelGGL(k_dec_result_init, dim3(1), dim3(1), 0, st, d_res);
    if (n_rsi == 0) return true;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const uint64_t n_items = n_rsi * c.segs_per_rsi;
    uint8_t *w = static_cast<uint8_t *>(d_ws);
    SegSum *sums = reinterpret_cast<SegSum *>(w);
    SegEntry *table = reinterpret_cast<SegEntry *>(w + up(n_items * sizeof(SegSum)));
    uint32_t *list = reinterpret_cast<uint32_t *>(w + up(n_items * sizeof(SegSum)) + up(n_items * sizeof(SegEntry)));
    uint32_t *list_cnt = list + ((up(n_rsi * 4)) / 4);
    (void)hipMemsetAsync(list_cnt, 0, 4, st);
    if (prof) (void)hipEventRecord(prof->ev[5], st);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(d_in);
    const uint64_t nwords = (in_bytes + 3) / 4;
    const uint64_t end_bit = (uint64_t)in_bytes * 8;
    BareArgs a_sum, a_seg, a_list;
    a_sum.sums = sums;
    a_list.list = list;
    a_list.list_cnt = list_cnt;
    a_sum.avg_hint = a_seg.avg_hint = a_list.avg_hint = avg_cds_hint;
    // (summing pass over the segments, scan per RSI, a lane per segment, a lane per RSI of the list)
    DecLaunch l{&c, words, nwords, end_bit, d_seg_bits, nullptr, n_items, total_blocks, nullptr, d_res, st, dump, d_idx, nullptr, 0u,
                a_sum};
    dec_part_bytes_bs(c.bs, true, true, l);
    hipLaunchKernelGGL(k_seg_scan, dim3((uint32_t)((n_rsi + 255) / 256)), dim3(256), 0, st, c, d_rsi_off, d_seg_bits,
                       sums, n_rsi, total_blocks, d_idx, table, list, list_cnt);
    l.rsi_off = nullptr;
    l.seg_table = table;
    l.out = d_out;
    l.ba = a_seg;
    dec_part_bytes_bs(c.bs, true, false, l);
    l.rsi_off = d_rsi_off;
    l.seg_table = nullptr;
    l.n_items = n_rsi;
    l.ba = a_list;
    dec_part_bytes_bs(c.bs, false, false, l);
    if (prof) (void)hipEventRecord(prof->ev[6], st);
    // (behind the timed kernels: return at once unless a k_decode raised kDecRedo)
    const uint64_t w_seg = (n_items + 63) / 64, w_rsi = (n_rsi + 63) / 64;
    hipLaunchKernelGGL((k_decode_redo<true>), dim3((uint32_t)(w_seg < 2048 ? w_seg : 2048)), dim3(64), 0, st, c, words,
                       nwords, end_bit, (const uint64_t *)nullptr, table, n_items, total_blocks, d_out, d_res, d_idx,
                       (const DecResult *)nullptr, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    hipLaunchKernelGGL((k_decode_redo<false>), dim3((uint32_t)(w_rsi < 2048 ? w_rsi : 2048)), dim3(64), 0, st, c, words,
                       nwords, end_bit, d_rsi_off, (const SegEntry *)nullptr, n_rsi, total_blocks, d_out, d_res, d_idx,
                       (const DecResult *)nullptr, 0u, list, list_cnt);
    return true;
}

bool launch_decode_segments(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const SegEntry *d_seg_table,
                            uint64_t n_seg, uint64_t total_blocks, uint8_t *d_out, DecResult *d_res,
                            hipStream_t st, const PhaseEvents *prof)
{
    return launch_decode_any<true>(c, d_in, in_bytes, nullptr, d_seg_table, n_seg, total_blocks, d_out, d_res, st,
                                   prof);
}
#endif      // AEC_DEC_PART

}  // namespace aec
