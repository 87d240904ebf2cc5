// aec_enc.hip -- gfx950 encoder kernels for the CCSDS 121.0-B-2 adaptive entropy coder.
//
// Work decomposition (MI355X: 64-lane wavefronts, 160 KiB LDS per CU):
//   segment  = up to 64 consecutive blocks of one RSI = one wavefront pass, ONE LANE PER BLOCK.
//              It is exactly the CCSDS zero-run segment (reference src/encode.c:649), so zero-block
//              aggregation is a 64-bit ballot per wavefront and never crosses a wavefront.
//   phase A  (lane = 16-byte chunk) coalesced global loads, byte-order fix-up, unit-delay
//            predictor + sign map (reference encode.c:235-311, encode_accessors.c:145-269),
//            results written to padded LDS rows [block][sample].
//   phase B  (lane = block) option / k-plateau / length selection (encode.c:313-434, 585-659).
//
//   k_analyze   phase A+B, writes a 4-byte summary per block and (bits, k clamp) per segment.
//   k_scan_*    device-wide exclusive scan of segment bit lengths (absolute bit offsets) and of
//               the k clamp composition (replaces the serial state->k / state->bits carry).
//               (k_scan_apply also zeroes the few output words that two waves of k_pack share)
//   k_pack      phase A again (input is re-read; the per-block summaries are not recomputed),
//               per-lane bit emission into an LDS image of the segment via ds_or_b32, then a
//               coalesced byte-swapped copy to HBM; only the first/last word of a segment can be
//               shared with a neighbour and uses a global atomic OR.
//
// HBM traffic per input byte: 2 reads of the input + 4/(bs*bytes) summary write+read
// + 1 write of the compressed stream.  Algorithmic bytes are N + C.
//
// Compiled once per templated block size with -DAEC_ENC_PART=<0|8|16|32|64> (k_analyze / k_pack / k_encode_fused of that
// block size: objects aec_enc_bs<N>.o) and once without (scans, batch kernels, dispatch), as aec_dec.hip is.
#include <hip/hip_runtime.h>

#include "aec_kernels.h"
#include "aec_lane.h"
#include "aec_tune.h"

// (grouped emission for 64-sample blocks: measured slower -- typical.dat shape 4.53 against 4.36 ms)
#ifndef AEC_ENC_GRP64
#define AEC_ENC_GRP64 0
#endif
namespace aec {

struct FusedGeom {
    uint32_t waves, segs, nparts, parts_per_wg, grid, obuf_words;
    size_t lds_bytes;
};
// The launches of one block size's kernels (defined in the object compiled with -DAEC_ENC_PART=BS)
template <int BS> void enc_part(bool pack, const Cfg &c, const uint8_t *in, const EncWorkspace &ws, uint32_t *out_words,
                                uint64_t cap_words, uint32_t fast_ok, hipStream_t st);
template <int BS> void enc_part_fused(const Cfg &c, const uint8_t *in, uint32_t *out_words, uint64_t cap_words,
                                      const FusedGeom &g, void *ctl, uint32_t start_bit, uint32_t k_in, uint64_t *rsi_off,
                                      SegEntry *seg_table, EncResult *res, uint32_t fast_ok, hipStream_t st);

namespace {

constexpr uint32_t kWave = 64;

// LDS rows [block][sample] of preprocessed samples.  Samples of at most 16 bits are kept as
// uint16 (half the LDS, which is what bounds the waves per CU of k_pack); wider ones as uint32.
// A row is padded by 16 bytes so that one-lane-per-row ds_read_b128 accesses are conflict free.
template <int BS, int BYTES>
struct Rows {
    static constexpr bool HALF = BS > 0 && (BYTES == 1 || BYTES == 2);
    __host__ __device__ static constexpr uint32_t stride_words(uint32_t bs) { return HALF ? bs / 2 + 4 : bs + 4; }
    __device__ static __forceinline__ void put(uint32_t *rows, uint32_t stride_w, uint32_t row, uint32_t col, uint32_t v)
    {
        if (HALF) reinterpret_cast<uint16_t *>(rows + row * stride_w)[col] = (uint16_t)v;
        else rows[row * stride_w + col] = v;
    }
};

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct LdsSink {
    uint32_t *w;
    __device__ __forceinline__ void or_word(uint32_t i, uint32_t v)
    {
        __hip_atomic_fetch_or(&w[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Wave-wide inclusive scan on the DPP network (row_shr within 16-lane rows, then row_bcast:15/31
// across rows): six cross-lane moves, no LDS traffic and no waits (the __shfl_up based version
// costs a dozen ds_bpermute round trips per scan).  op(earlier, later) may be non-commutative.
template <class Op>
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v, uint32_t ident, Op op)
{
    uint32_t t;
    t = __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false); v = op(t, v);   // row_shr:1
    t = __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false); v = op(t, v);   // row_shr:2
    t = __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false); v = op(t, v);   // row_shr:4
    t = __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false); v = op(t, v);   // row_shr:8
    t = __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false); v = op(t, v);   // row_bcast:15
    t = __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false); v = op(t, v);   // row_bcast:31
    return v;
}
// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);               // wave_shr:1
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return __builtin_amdgcn_readlane(v, 63); }

__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v, uint32_t)
{
    return wave_scan_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}

// ordered inclusive scan of k clamps packed as lo | hi << 8
__device__ __forceinline__ uint32_t clamp_pack(KClamp c) { return c.lo | (c.hi << 8); }
__device__ __forceinline__ KClamp clamp_unpack(uint32_t p) { return KClamp{p & 0xFFu, (p >> 8) & 0xFFu}; }

// Inclusive scan of clamp compositions ("first a, then b": both bounds of a are clamped by b).
// Inside the scan a clamp is a pair of 16-bit halves (lo, hi), so one step is two byte permutes
// that replicate b's bounds and a packed max and min -- instead of unpacking, four compares /
// selects and repacking.  In and out: the lo | hi << 8 form of clamp_pack.
__device__ __forceinline__ uint32_t wave_incl_clamp(uint32_t v, uint32_t)
{
    typedef unsigned short cl16x2 __attribute__((ext_vector_type(2)));
    const uint32_t wide = (v & 0xFFu) | ((v & 0xFF00u) << 8);
    const KClamp id = kclamp_identity();
    const uint32_t r = wave_scan_dpp(wide, id.lo | (id.hi << 16), [](uint32_t a, uint32_t b) {
        const uint32_t blo = __builtin_amdgcn_perm(b, b, 0x01000100u), bhi = __builtin_amdgcn_perm(b, b, 0x03020302u);
        const cl16x2 t = __builtin_elementwise_min(
            __builtin_elementwise_max(__builtin_bit_cast(cl16x2, a), __builtin_bit_cast(cl16x2, blo)),
            __builtin_bit_cast(cl16x2, bhi));
        return __builtin_bit_cast(uint32_t, t);
    });
    return (r & 0xFFu) | ((r >> 8) & 0xFF00u);
}

struct Seg {
    uint64_t rsi_idx;   // RSI this segment belongs to
    uint64_t blk0;      // global index of its first block
    uint64_t samp0;     // global index of its first sample
    uint32_t b0;        // index of its first block inside the RSI (multiple of 64)
    uint32_t nv;        // blocks in the segment (1..64)
    bool full;          // every sample of the segment exists (no end-of-data padding)
};

// everything that follows from (RSI, segment inside the RSI)
__device__ __forceinline__ Seg seg_at(const Cfg &c, uint64_t rsi_idx, uint32_t s);

__device__ __forceinline__ Seg seg_geom(const Cfg &c, uint64_t sg)
{
    // 32-bit division whenever possible (a 64-bit one costs ~100 instructions per segment)
    const uint64_t rsi_idx = (sg >> 32) ? sg / c.segs_per_rsi : (uint64_t)((uint32_t)sg / c.segs_per_rsi);
    return seg_at(c, rsi_idx, (uint32_t)(sg - rsi_idx * c.segs_per_rsi));
}

// the segment after g: a wave walks consecutive segments, so the division is paid once per wave
__device__ __forceinline__ Seg seg_next(const Cfg &c, const Seg &g)
{
    const uint32_t s = g.b0 / 64u + 1u;
    return s == c.segs_per_rsi ? seg_at(c, g.rsi_idx + 1, 0u) : seg_at(c, g.rsi_idx, s);
}

__device__ __forceinline__ Seg seg_at(const Cfg &c, uint64_t rsi_idx, uint32_t s)
{
    Seg g;
    g.rsi_idx = rsi_idx;
    g.b0 = s * 64u;
    uint64_t nb = c.total_blocks - g.rsi_idx * c.rsi;
    if (nb > c.rsi) nb = c.rsi;
    const uint64_t left = nb - g.b0;
    g.nv = left < 64 ? (uint32_t)left : 64u;
    g.blk0 = g.rsi_idx * c.rsi + g.b0;
    g.samp0 = g.blk0 * c.bs;
    // "full": every sample exists and the segment is a whole number of 16-byte chunks (a short
    // final RSI may end on half a chunk)
    g.full = g.samp0 + (uint64_t)g.nv * c.bs <= c.total_samples &&
             ((uint64_t)g.nv * c.bs * c.bytes) % 16 == 0;
    return g;
}

// ---- phase A, fast path: whole 16-byte chunks, segment 16-byte aligned in HBM ----------------
// Split into ISSUE (global loads into registers) and FINISH (byte order, predictor, LDS rows) so
// that the loads of segment i+1 are in flight while segment i is analysed / packed: a wavefront
// otherwise exposes a full HBM round trip several times per segment.
template <int BYTES>
__device__ __forceinline__ uint32_t load_sample_raw(const uint8_t *p)
{
    if (BYTES == 1) return *p;
    if (BYTES == 2) return *reinterpret_cast<const uint16_t *>(p);
    return *reinterpret_cast<const uint32_t *>(p);
}

// byte order of a sample loaded with load_sample_raw -- applied where the value is CONSUMED: done
// at the load it makes the prefetch wait for its own data (a full HBM round trip per segment)
template <int BYTES>
__device__ __forceinline__ uint32_t sample_byte_order(uint32_t v, bool msb)
{
    if (BYTES == 1) return v;
    if (BYTES == 2) return msb ? ((v >> 8) | ((v & 0xFFu) << 8)) : v;
    return msb ? bswap32(v) : v;
}

// ---- two 16-bit samples per instruction ------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t unpk(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// The unsigned predictor mapping (reference encode.c:255-269; aec_lane.h pp_unsigned) for both
// halves of a word: prev = the two preceding samples, cur = the two samples, xm = (xmax, xmax).
// With D = |cur - prev|, neg = cur < prev and room = min(prev, xmax - prev) the mapped value is
// 2D - neg while D <= room and room + D beyond; the first grows twice as fast as the second and
// they cross exactly at the switch, so it is min(2D - neg, room + D) -- no compare, no select,
// and everything stays below 2^16 (the doubling saturates, which the min absorbs).
__device__ __forceinline__ uint32_t pp_unsigned_pk(uint32_t prev, uint32_t cur, uint32_t xm)
{
    const u16x2 a = pk(prev), b = pk(cur), one = {1, 1};
    const u16x2 d = __builtin_elementwise_max(a, b) - __builtin_elementwise_min(a, b);
    const u16x2 neg = __builtin_elementwise_min(__builtin_elementwise_sub_sat(a, b), one);
    const u16x2 room = __builtin_elementwise_min(a, pk(xm) - a);
    const u16x2 folded = __builtin_elementwise_add_sat(d, d - neg);
    return unpk(__builtin_elementwise_min(folded, room + d));
}

// The mapped residuals of NW words of sample pairs in stream order (pw; `before` holds the sample in front of them in its
// upper half).  Where nothing can clip -- every step of the stretch is at most the distance between its smallest and its
// largest sample, R, and R fits the room between those and the ends of the range: one test per stretch, wave-wide -- a
// residual is the zigzag of the difference (subtract, shift, sign, exclusive or: 4 packed instructions per pair against
// 11 for pp_unsigned_pk, which the wavefronts with a lane near the ends of the range run as before).  |difference| <= R <=
// xmax / 2 < 2^15, so the difference is exact as a signed 16-bit value and twice it fits 16 bits.
// (try_fast: wave-uniform, the caller's; a stretch that does not pass ends the tests for the caller's remaining stretches
// -- data that lives near an end of its range pays one test per segment, not one per stretch)
template <uint32_t NW>
__device__ __forceinline__ void pp_words_pk(const uint32_t *pw, uint32_t before, uint32_t xm, bool act, uint32_t *out,
                                            bool &try_fast)
{
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    if (!try_fast) {
#pragma unroll
        for (uint32_t j = 0; j < NW; j++)
            out[j] = pp_unsigned_pk(__builtin_amdgcn_alignbit(pw[j], j ? pw[j - 1] : before, 16), pw[j], xm);
        return;
    }
    u16x2 lo = pk((before >> 16) * 0x00010001u), hi = lo;
#pragma unroll
    for (uint32_t j = 0; j < NW; j++) {
        lo = __builtin_elementwise_min(lo, pk(pw[j]));
        hi = __builtin_elementwise_max(hi, pk(pw[j]));
    }
    const uint32_t lo_s = lo.x < lo.y ? lo.x : lo.y, hi_s = hi.x > hi.y ? hi.x : hi.y, range = hi_s - lo_s;
    const bool fits = range <= lo_s && range <= (xm & 0xFFFFu) - hi_s;
    if (!__any(act && !fits)) {
        const u16x2 one = {1, 1}, fifteen = {15, 15};
#pragma unroll
        for (uint32_t j = 0; j < NW; j++) {
            const u16x2 d = pk(pw[j]) - pk(__builtin_amdgcn_alignbit(pw[j], j ? pw[j - 1] : before, 16));
            const i16x2 sign = __builtin_bit_cast(i16x2, d) >> __builtin_bit_cast(i16x2, fifteen);
            out[j] = unpk(d << one) ^ (uint32_t)__builtin_bit_cast(uint32_t, sign);
        }
        return;
    }
    try_fast = false;
#pragma unroll
    for (uint32_t j = 0; j < NW; j++)
        out[j] = pp_unsigned_pk(__builtin_amdgcn_alignbit(pw[j], j ? pw[j - 1] : before, 16), pw[j], xm);
}

template <int BS, int BYTES>
struct FastSeg {
    static constexpr uint32_t CHUNKS = (uint32_t)BS * BYTES * 64u / 16u;       // per full segment
    static constexpr uint32_t NIT = (CHUNKS + 63u) / 64u;                      // chunk rounds per lane
    uint4 v[NIT];
    uint32_t carry;   // sample just before the segment, as loaded (byte order not yet resolved)
};

// max_chunk = index of the last whole 16-byte chunk of the input (addresses are clamped to it, so
// the loads are unconditional and can be issued for a segment that later takes the generic path)
template <int BS, int BYTES>
__device__ __forceinline__ void fast_issue(const Cfg &c, const uint8_t *in, const Seg &g, uint32_t lane,
                                           uint64_t max_chunk, FastSeg<BS, BYTES> &f)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(in);
    const uint64_t first = g.samp0 * BYTES / 16u;
#pragma unroll
    for (uint32_t it = 0; it < FastSeg<BS, BYTES>::NIT; it++) {
        uint64_t ci = first + it * kWave + lane;
        if (ci > max_chunk) ci = max_chunk;
        f.v[it] = src[ci];
    }
    const uint64_t prev = g.samp0 ? g.samp0 - 1 : 0;
    f.carry = load_sample_raw<BYTES>(in + prev * BYTES);
}

template <int BS, int BYTES>
__device__ __forceinline__ void fast_finish(const Cfg &c, const Seg &g, const FastSeg<BS, BYTES> &f,
                                            uint32_t *rows, uint32_t lane)
{
    constexpr uint32_t SPC = 16 / BYTES;           // samples per chunk
    constexpr uint32_t STRIDE = Rows<BS, BYTES>::stride_words(BS);
    const bool msb = c.flags & F_MSB, pp = c.flags & F_PREPROCESS;
    const uint32_t nchunks = g.nv * (uint32_t)BS * BYTES / 16u;
    uint32_t carry = sample_byte_order<BYTES>(f.carry, msb);
    bool try_fast = true;                          // (pp_words_pk)

#pragma unroll
    for (uint32_t it = 0; it < FastSeg<BS, BYTES>::NIT; it++) {
        const uint32_t ci = it * kWave + lane;
        const bool act = ci < nchunks;
        const uint4 v = f.v[it];
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
        if (Rows<BS, BYTES>::HALF && pp) {
            // samples of at most 16 bits stay packed, two per instruction, from the loaded words to the
            // uint16 rows (pw[] = sample pairs in stream order).  Signed samples are biased by
            // 2^(bps-1) first -- flip the sign bit, drop the sign copies above it -- which turns
            // [xmin, xmax] into [0, 2^bps - 1] and leaves differences and distances to the bounds,
            // hence the mapped values (reference encode.c:294-309), unchanged.
            constexpr uint32_t NW = SPC / 2;
            uint32_t pw[NW];
            if (BYTES == 2) {
#pragma unroll
                for (uint32_t j = 0; j < 4; j++)
                    pw[j] = msb ? (((vw[j] & 0x00FF00FFu) << 8) | ((vw[j] >> 8) & 0x00FF00FFu)) : vw[j];
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {      // bytes 0,1 and 2,3 widened to 16 bits each
                    pw[2 * j] = __builtin_amdgcn_perm(0u, vw[j], 0x0c010c00u);
                    pw[2 * j + 1] = __builtin_amdgcn_perm(0u, vw[j], 0x0c030c02u);
                }
            }
            const bool sgn = c.flags & F_SIGNED;
            const uint32_t full = low_mask32(c.bps);
            if (sgn) {
                const uint32_t flip = (1u << (c.bps - 1)) * 0x00010001u, keep = full * 0x00010001u;
#pragma unroll
                for (uint32_t j = 0; j < NW; j++) pw[j] = (pw[j] ^ flip) & keep;
            }
            // the word before this chunk: the neighbour lane's last word, or the carried sample
            const uint32_t cbias = sgn ? (carry ^ (1u << (c.bps - 1))) & full : carry;
            const uint32_t before = wave_shr1(pw[NW - 1], cbias << 16);
            carry = wave_last(pw[NW - 1]) >> 16;
            if (sgn) carry ^= 1u << (c.bps - 1);       // back to the raw form the next round expects
            const uint32_t xm = (sgn ? full : c.xmax) * 0x00010001u;
            uint32_t dw[NW];
            pp_words_pk<NW>(pw, before, xm, act, dw, try_fast);
            if (act) {
                if (g.b0 == 0 && ci == 0) dw[0] &= 0xFFFF0000u;      // reference sample slot, encode.c:254
#pragma unroll
                for (uint32_t q = 0; q < NW / 4; q++) {
                    const uint32_t i = ci * SPC + q * 8;
                    const uint32_t row = i / BS, col = i % BS;
                    *reinterpret_cast<uint4 *>(&rows[row * STRIDE + col / 2]) =
                        make_uint4(dw[4 * q], dw[4 * q + 1], dw[4 * q + 2], dw[4 * q + 3]);
                }
            }
            continue;
        }
        uint32_t x[SPC];
        if (BYTES == 4) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) x[j] = msb ? bswap32(vw[j]) : vw[j];
        } else if (BYTES == 2) {
            uint32_t sw[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)   // swap the bytes of both halves at once for MSB data
                sw[j] = msb ? (((vw[j] & 0x00FF00FFu) << 8) | ((vw[j] >> 8) & 0x00FF00FFu)) : vw[j];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) x[j] = (j & 1) ? sw[j >> 1] >> 16 : sw[j >> 1] & 0xFFFFu;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) x[j] = (vw[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        }
        const uint32_t last = x[SPC - 1];
        const uint32_t prev = wave_shr1(last, carry);
        carry = wave_last(last);
        if (act) {
            uint32_t dd[SPC];
            // the three flavours are selected once per chunk (wave-uniform), not per sample
            if (!pp) {
#pragma unroll
                for (uint32_t j = 0; j < SPC; j++) dd[j] = x[j];
            } else {
                // as pp_words_pk, for samples of more than 16 bits: the values as distances from xmin, their smallest and
                // largest, the zigzag of the differences where nothing can clip (|difference| <= the stretch's range <=
                // half the sample range < 2^31), the exact mapping otherwise -- and then for the rest of the segment
                const bool sgn = c.flags & F_SIGNED;
                bool fast = try_fast;
                uint32_t u[SPC];
                const uint32_t up = sgn ? sign_extend(prev, c.bps) - c.xmin : prev;
                if (fast) {
                    uint32_t lo = up, hi = up;
#pragma unroll
                    for (uint32_t j = 0; j < SPC; j++) {
                        u[j] = sgn ? sign_extend(x[j], c.bps) - c.xmin : x[j];
                        lo = u[j] < lo ? u[j] : lo;
                        hi = u[j] > hi ? u[j] : hi;
                    }
                    const uint32_t range = hi - lo;
                    fast = !__any(!(range <= lo && range <= (c.xmax - c.xmin) - hi));
                    try_fast = fast;
                }
                if (fast) {
#pragma unroll
                    for (uint32_t j = 0; j < SPC; j++) {
                        const uint32_t diff = u[j] - (j ? u[j - 1] : up);
                        dd[j] = (diff << 1) ^ (uint32_t)((int32_t)diff >> 31);
                    }
                } else if (sgn) {
                    uint32_t pv = sign_extend(prev, c.bps);
#pragma unroll
                    for (uint32_t j = 0; j < SPC; j++) {
                        const uint32_t cv = sign_extend(x[j], c.bps);
                        dd[j] = pp_signed(pv, cv, c.xmin, c.xmax);
                        pv = cv;
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < SPC; j++) dd[j] = pp_unsigned(j ? x[j - 1] : prev, x[j], c.xmax);
                }
            }
            if (pp && g.b0 == 0 && ci == 0) dd[0] = 0;   // reference sample slot, encode.c:254
            if (Rows<BS, BYTES>::HALF) {
#pragma unroll
                for (uint32_t q = 0; q < SPC / 8; q++) {
                    const uint32_t i = ci * SPC + q * 8;
                    const uint32_t row = i / BS, col = i % BS;
                    *reinterpret_cast<uint4 *>(&rows[row * STRIDE + col / 2]) =
                        make_uint4(dd[8 * q] | (dd[8 * q + 1] << 16), dd[8 * q + 2] | (dd[8 * q + 3] << 16),
                                   dd[8 * q + 4] | (dd[8 * q + 5] << 16), dd[8 * q + 6] | (dd[8 * q + 7] << 16));
                }
            } else {
#pragma unroll
                for (uint32_t q = 0; q < SPC / 4; q++) {
                    const uint32_t i = ci * SPC + q * 4;
                    const uint32_t row = i / BS, col = i % BS;
                    *reinterpret_cast<uint4 *>(&rows[row * STRIDE + col]) =
                        make_uint4(dd[4 * q], dd[4 * q + 1], dd[4 * q + 2], dd[4 * q + 3]);
                }
            }
        }
    }
}

// ---- phase A, direct path: blocks of at most 32 bytes of samples of at most 16 bits ------------------
// Every lane loads ITS block (lane = block from the start: no chunk -> row transposition through the LDS),
// maps it with the packed predictor in its own registers -- the sample before the block comes from the
// neighbour lane -- and hands the sample pairs to the analysis / emission as they are.  The differential
// profile (profiles/r02/differential_profile_c2.txt) put the LDS feed at 0.69 + 0.78 ms of the 3.6 ms the two
// encoder kernels take at C2; most of it was not the predictor but the way through the rows.
template <int BS, int BYTES>
struct DirectSeg {
    static constexpr uint32_t NRAW = (uint32_t)BS * BYTES / 4u;      // 2, 4 or 8 words per block
    uint32_t raw[NRAW];
    uint32_t carry;   // sample just before the segment, as loaded
};

// max_blk = index of the last whole block of the input (lanes beyond the segment re-read a valid block)
template <int BS, int BYTES>
__device__ __forceinline__ void direct_issue(const Cfg &c, const uint8_t *in, const Seg &g, uint32_t lane,
                                             uint64_t max_blk, DirectSeg<BS, BYTES> &f)
{
    constexpr uint32_t BLK = (uint32_t)BS * BYTES;
    uint64_t bi = g.blk0 + lane;
    if (bi > max_blk) bi = max_blk;
    const uint8_t *p = in + bi * BLK;
    if (BLK == 8) {
        const uint2 a = *reinterpret_cast<const uint2 *>(p);
        f.raw[0] = a.x; f.raw[1] = a.y;
    } else {
#pragma unroll
        for (uint32_t q = 0; q < BLK / 16u; q++) {
            const uint4 a = reinterpret_cast<const uint4 *>(p)[q];
            f.raw[4 * q] = a.x; f.raw[4 * q + 1] = a.y; f.raw[4 * q + 2] = a.z; f.raw[4 * q + 3] = a.w;
        }
    }
    const uint64_t prev = g.samp0 ? g.samp0 - 1 : 0;
    f.carry = load_sample_raw<BYTES>(in + prev * BYTES);
}

// the lane's block as BS / 2 words of two mapped 16-bit samples each (what the uint16 rows hold)
template <int BS, int BYTES>
__device__ __forceinline__ void direct_finish(const Cfg &c, const Seg &g, const DirectSeg<BS, BYTES> &f,
                                              uint32_t lane, uint32_t *w)
{
    constexpr uint32_t NW = (uint32_t)BS / 2u;
    const bool msb = c.flags & F_MSB, sgn = c.flags & F_SIGNED;
    uint32_t pw[NW];
    if (BYTES == 2) {
#pragma unroll
        for (uint32_t j = 0; j < NW; j++)
            pw[j] = msb ? (((f.raw[j] & 0x00FF00FFu) << 8) | ((f.raw[j] >> 8) & 0x00FF00FFu)) : f.raw[j];
    } else {
#pragma unroll
        for (uint32_t j = 0; j < NW / 2u; j++) {      // bytes 0,1 and 2,3 widened to 16 bits each
            pw[2 * j] = __builtin_amdgcn_perm(0u, f.raw[j], 0x0c010c00u);
            pw[2 * j + 1] = __builtin_amdgcn_perm(0u, f.raw[j], 0x0c030c02u);
        }
    }
    // signed samples biased by 2^(bps-1) as in fast_finish
    const uint32_t full = low_mask32(c.bps);
    uint32_t carry = sample_byte_order<BYTES>(f.carry, msb);
    if (sgn) {
        const uint32_t flip = (1u << (c.bps - 1)) * 0x00010001u, keep = full * 0x00010001u;
#pragma unroll
        for (uint32_t j = 0; j < NW; j++) pw[j] = (pw[j] ^ flip) & keep;
        carry = (carry ^ (1u << (c.bps - 1))) & full;
    }
    const uint32_t before = wave_shr1(pw[NW - 1], carry << 16);
    const uint32_t xm = (sgn ? full : c.xmax) * 0x00010001u;
    bool try_fast = true;
    pp_words_pk<NW>(pw, before, xm, lane < g.nv, w, try_fast);
    if (g.b0 == 0 && lane == 0) w[0] &= 0xFFFF0000u;      // reference sample slot, encode.c:254
}

// ---- phase A, generic path: lane = sample, byte loads, end-of-data padding -------------------
template <int BS, int BYTES>
__device__ __forceinline__ void load_segment_generic(const Cfg &c, const uint8_t *in, const Seg &g,
                                                     uint32_t *rows, uint32_t stride, uint32_t lane)
{
    const bool msb = c.flags & F_MSB, pp = c.flags & F_PREPROCESS;
    const uint32_t ns = g.nv * c.bs;
    const uint64_t last = c.total_samples - 1;
    for (uint32_t i = lane; i < ns; i += kWave) {
        const uint64_t gi = g.samp0 + i;
        const uint64_t gc = gi < last ? gi : last;            // encode.c:676-684: repeat last sample
        const uint32_t cur = load_sample_bytes(in + gc * c.bytes, c.bytes, msb);
        uint32_t dv;
        if (!pp) {
            dv = cur;
        } else if (g.b0 == 0 && i == 0) {
            dv = 0;
        } else {
            const uint64_t gp = gi - 1 < last ? gi - 1 : last;
            dv = pp_any(load_sample_bytes(in + gp * c.bytes, c.bytes, msb), cur, c);
        }
        Rows<BS, BYTES>::put(rows, stride, i / c.bs, i % c.bs, dv);
    }
}

// Segment feeder: owns the prefetched registers of the NEXT segment.  fast == templated block
// size, 1/2/4-byte containers, 16-byte aligned RSIs; everything else takes the generic loader.
template <int BS, int BYTES>
struct Feeder {
    static constexpr bool FAST_T = BS > 0 && (BYTES == 1 || BYTES == 2 || BYTES == 4);
    static constexpr int FBS = FAST_T ? BS : 8, FBY = FAST_T ? BYTES : 1;
    // prefetch across segments only while it is cheap in registers (<= 4 x 16 bytes per lane)
    static constexpr bool PIPE = FAST_T && FastSeg<FBS, FBY>::NIT <= 4;
    FastSeg<(PIPE ? FBS : 8), (PIPE ? FBY : 1)> pre;
    bool fast;            // run-time half of the decision (alignment, at least one whole chunk)
    uint64_t max_chunk;

    __device__ __forceinline__ void init(const Cfg &c, uint32_t fast_ok)
    {
        const uint64_t total_bytes = c.total_samples * c.bytes;
        fast = FAST_T && fast_ok && total_bytes >= 16;
        max_chunk = total_bytes >= 16 ? total_bytes / 16 - 1 : 0;
        const uint64_t whole_blocks = c.total_samples / (BS ? (uint32_t)BS : c.bs);
        max_blk = whole_blocks ? whole_blocks - 1 : 0;
        if (whole_blocks == 0) fast = false;
    }
    __device__ __forceinline__ void prefetch(const Cfg &c, const uint8_t *in, const Seg &g, uint32_t lane)
    {
        if (PIPE) {
            if (fast) fast_issue<(PIPE ? FBS : 8), (PIPE ? FBY : 1)>(c, in, g, lane, max_chunk, pre);
        }
    }
    // ---- direct path (see DirectSeg): kernels that take it prefetch with prefetch_direct and fall back to
    // feed_now for the segments it does not cover (end-of-data padding, no preprocessing)
    static constexpr bool DIRECT = FAST_T && Rows<BS, BYTES>::HALF && BS * BYTES <= 32;
    DirectSeg<(DIRECT ? BS : 8), (DIRECT ? BYTES : 1)> pre_direct;
    uint64_t max_blk;
    __device__ __forceinline__ bool direct_ok(const Cfg &c, const Seg &g) const
    {
        return DIRECT && fast && g.full && (c.flags & F_PREPROCESS);
    }
    __device__ __forceinline__ void prefetch_direct(const Cfg &c, const uint8_t *in, const Seg &g, uint32_t lane)
    {
        if (DIRECT) {
            if (fast) direct_issue<(DIRECT ? BS : 8), (DIRECT ? BYTES : 1)>(c, in, g, lane, max_blk, pre_direct);
        }
    }
    // rows of g without a prefetch (fallback of the direct kernels)
    __device__ __forceinline__ void feed_now(const Cfg &c, const uint8_t *in, const Seg &g, uint32_t *rows,
                                             uint32_t stride, uint32_t lane)
    {
        if (FAST_T) {
            if (fast && g.full) {
                FastSeg<FBS, FBY> now;
                fast_issue<FBS, FBY>(c, in, g, lane, max_chunk, now);
                fast_finish<FBS, FBY>(c, g, now, rows, lane);
                return;
            }
        }
        load_segment_generic<BS, BYTES>(c, in, g, rows, stride, lane);
    }
    // consumes the registers prefetched for g (call prefetch(next) BEFORE this to overlap)
    __device__ __forceinline__ void feed(const Cfg &c, const uint8_t *in, const Seg &g,
                                         const FastSeg<(PIPE ? FBS : 8), (PIPE ? FBY : 1)> &cur, uint32_t *rows,
                                         uint32_t stride, uint32_t lane)
    {
        if (PIPE) {
            if (fast && g.full) {
                fast_finish<(PIPE ? FBS : 8), (PIPE ? FBY : 1)>(c, g, cur, rows, lane);
                return;
            }
        } else if (FAST_T) {
            if (fast && g.full) {          // big blocks: load and use in place
                FastSeg<FBS, FBY> now;
                fast_issue<FBS, FBY>(c, in, g, lane, max_chunk, now);
                fast_finish<FBS, FBY>(c, g, now, rows, lane);
                return;
            }
        }
        load_segment_generic<BS, BYTES>(c, in, g, rows, stride, lane);
    }
};

// copy a block's LDS row into registers (BS > 0) or alias the row (BS == 0)
template <int BS, bool HALF>
struct BlockRegs {
    uint32_t v[BS];
    __device__ __forceinline__ const uint32_t *load(const uint32_t *row)
    {
        if (HALF) {
#pragma unroll
            for (int q = 0; q < BS / 8; q++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(row + 4 * q);
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[8 * q + 2 * j] = w[j] & 0xFFFFu;
                    v[8 * q + 2 * j + 1] = w[j] >> 16;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < BS / 4; q++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(row + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        }
        return v;
    }
};
template <bool HALF>
struct BlockRegs<0, HALF> {
    __device__ __forceinline__ const uint32_t *load(const uint32_t *row) { return row; }
};

template <int BS>
__device__ __forceinline__ bool block_is_zero(const uint32_t *d, uint32_t bs_rt)
{
    const uint32_t bs = BS ? (uint32_t)BS : bs_rt;
    uint32_t any = 0;
#pragma unroll
    for (uint32_t i = 0; i < bs; i++) any |= d[i];
    return any == 0;
}

// Option selection on a block held as BS/2 words of two 16-bit residuals each (uint16 rows):
// fs(k) = sum of (sample >> k) costs one packed shift and one dot product per PAIR
// (v_pk_lshrrev_b16, v_dot2_u32_u16 with (1, 1)); the control flow is aec_lane.h's.
template <int BS>
__device__ __forceinline__ BlockChoice choose_option_pk(const uint32_t *w, const Cfg &c, uint32_t ref)
{
    const uint32_t n = (uint32_t)BS - ref;
    const u16x2 ones = {1, 1};
    auto fs = [&](uint32_t k) -> uint32_t {       // at most 64 * 65535 < 2^22
        const u16x2 kk = {(unsigned short)k, (unsigned short)k};
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < BS / 2; j++) s = __builtin_amdgcn_udot2(pk(w[j]) >> kk, ones, s, false);
        return s;
    };
    uint32_t split_len = 0xFFFFFFFFu, klo = 0, khi = 31;
    if (c.id_len > 1) assess_split_with(fs, n, c.kmax, klo, khi, split_len);
    // second extension (aec_lane.h assess_se): a pair is the two halves of a word; their sum stays
    // below 2^17, so the reference's 64-bit wrap case cannot occur here
    // (round 6) The reference adds (a + b)(a + b + 1) / 2 + b + 1 pair by pair and gives up once the length passes the
    // limit (encode.c:412-434); the terms are positive, so that is "the whole sum passes the limit", and the whole sum is
    // 1 + (sum of s^2 + sum of s) / 2 + sum of b + pairs with s = a + b: per pair two dot products, a multiply-add, an or
    // and an add instead of a dozen instructions with their compares and selects.  A pair sum of 2^13 and more is beyond
    // any limit (<= 64 x 16 bits) by itself -- the or of all pair sums tells -- and below that nothing overflows 32 bits.
    const uint32_t limit = n * c.bps;
    const u16x2 second = {0, 1};
    uint32_t s_all = 0, s_sq = 0, s_b = 0, s_or = 0;
#pragma unroll
    for (int j = 0; j < BS / 2; j++) {
        const uint32_t sum = __builtin_amdgcn_udot2(pk(w[j]), ones, 0u, false);
        s_or |= sum;
        s_all += sum;
        s_sq += __umul24(sum, sum);                                   // (sum < 2^13 wherever the result is used)
        s_b = __builtin_amdgcn_udot2(pk(w[j]), second, s_b, false);
    }
    const uint32_t len = 1u + (s_sq + s_all) / 2u + s_b + (uint32_t)BS / 2u;
    const bool over = (s_or >> 13) != 0u || len > limit;
    return choose_from(c, (uint32_t)BS, ref, split_len, over ? 0xFFFFFFFFu : len, klo, khi);
}

// ----------------------------------------------------------------------------------------------
// K1: analysis
// ----------------------------------------------------------------------------------------------
template <int BS, int BYTES>
__device__ __forceinline__ void analyze_body(const Cfg &c, const Seg &g, uint32_t *rows, uint32_t stride,
                                             uint32_t lane, uint64_t sg, uint32_t *__restrict__ meta,
                                             uint32_t *__restrict__ seg_bits, uint16_t *__restrict__ seg_clamp,
                                             const uint32_t *direct = nullptr);

template <int BS, int BYTES>
__global__ void __launch_bounds__(256)
k_analyze(const Cfg c, const uint8_t *__restrict__ in, uint32_t *__restrict__ meta,
          uint32_t *__restrict__ seg_bits, uint16_t *__restrict__ seg_clamp, uint32_t segs_per_wave,
          uint32_t fast_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: segment geometry and addresses then run on the SALU)
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const uint32_t stride = Rows<BS, BYTES>::stride_words(bs);
    uint32_t *rows = smem + (size_t)wave * 64u * stride;

    const uint64_t gwave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    uint64_t sg = gwave * segs_per_wave;
    uint64_t sg_end = sg + segs_per_wave;
    if (sg_end > c.total_segs) sg_end = c.total_segs;

    Feeder<BS, BYTES> feeder;
    feeder.init(c, fast_ok);
    Seg g = seg_geom(c, sg < sg_end ? sg : 0);
    if (Feeder<BS, BYTES>::DIRECT) {
        // small blocks: lane = block from the load on, nothing goes through the rows (Feeder::DIRECT)
        if (sg < sg_end) feeder.prefetch_direct(c, in, g, lane);
        for (; sg < sg_end; sg++) {
            const auto cur = feeder.pre_direct;
            const Seg gcur = g;
            if (sg + 1 < sg_end) g = seg_next(c, g);
            feeder.prefetch_direct(c, in, g, lane);       // the next segment's loads fly during this one
            if (feeder.direct_ok(c, gcur)) {
                uint32_t w[BS ? BS / 2 : 1];
                direct_finish<(Feeder<BS, BYTES>::DIRECT ? BS : 8), (Feeder<BS, BYTES>::DIRECT ? BYTES : 1)>(c, gcur, cur, lane, w);
                analyze_body<BS, BYTES>(c, gcur, rows, stride, lane, sg, meta, seg_bits, seg_clamp, w);
            } else {
                feeder.feed_now(c, in, gcur, rows, stride, lane);
                analyze_body<BS, BYTES>(c, gcur, rows, stride, lane, sg, meta, seg_bits, seg_clamp);
            }
        }
        return;
    }
    if (sg < sg_end) feeder.prefetch(c, in, g, lane);
    for (; sg < sg_end; sg++) {
        const auto cur = feeder.pre;
        const Seg gcur = g;
        if (sg + 1 < sg_end) g = seg_next(c, g);
        feeder.prefetch(c, in, g, lane);          // next segment's loads fly during this one
        feeder.feed(c, in, gcur, cur, rows, stride, lane);
        analyze_body<BS, BYTES>(c, gcur, rows, stride, lane, sg, meta, seg_bits, seg_clamp);
    }
}

// Phase B of one segment (rows already in LDS): this lane's block summary (meta_pack), the segment's
// bit length and its k clamp (lo | hi << 8), both wave-uniform.
template <int BS, int BYTES>
__device__ __forceinline__ uint32_t analyze_segment(const Cfg &c, const Seg &g, const uint32_t *rows, uint32_t stride,
                                                    uint32_t lane, uint32_t &tot, uint32_t &cl,
                                                    const uint32_t *direct = nullptr)   // the lane's block from direct_finish
{
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const bool pp = c.flags & F_PREPROCESS;
    constexpr bool WIDE_T = (BYTES >= 3);
    BlockRegs<BS, Rows<BS, BYTES>::HALF && BS != 0 ? false : Rows<BS, BYTES>::HALF> regs;
    constexpr bool PK = Rows<BS, BYTES>::HALF && BS != 0;   // uint16 rows: analysed two samples at a time
    const bool valid = lane < g.nv;
    // every lane loads a row (idle lanes re-read the last valid one) so that the block lives in
    // registers instead of behind a conditionally assigned pointer
    const uint32_t *row = rows + (valid ? lane : g.nv - 1) * stride;
    uint32_t w[PK ? BS / 2 : 1];                            // the block as sample pairs
    const uint32_t *d = nullptr;
    bool zero;
    if (PK) {
        if (direct) {
#pragma unroll
            for (int j = 0; j < (PK ? BS / 2 : 0); j++) w[j] = direct[j];
        } else {
#pragma unroll
            for (int q = 0; q < (PK ? BS / 8 : 0); q++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(row + 4 * q);
                w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
            }
        }
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < (PK ? BS / 2 : 0); j++) any |= w[j];
        zero = valid && any == 0;
    } else {
        d = regs.load(row);
        zero = valid && block_is_zero<BS>(d, bs);
    }
    const uint64_t zmask = __ballot(zero);

    uint32_t m = meta_pack(0, OPT_ZCONT, 0, 0);
    KClamp kc = kclamp_identity();
    if (valid) {
        const uint32_t ref = (pp && g.b0 == 0 && lane == 0) ? 1u : 0u;
        if (zero) {
            uint32_t fs = 0;
            const uint32_t run = zero_run_at(zmask, lane, g.nv, fs);
            if (run) m = meta_pack(c.id_len + 1 + ref * c.bps + fs + 1, OPT_ZERO, fs, 0);
        } else {
            BlockChoice ch;
            if (PK)
                ch = choose_option_pk<(PK ? BS : 8)>(w, c, ref);
            else if (BS == 0)
                ch = (c.bps > 16) ? choose_option<BS, true>(d, c, ref) : choose_option<BS, false>(d, c, ref);
            else
                ch = choose_option<BS, WIDE_T>(d, c, ref);
            m = meta_pack(ch.bits, ch.opt, ch.klo, ch.khi);
            if (c.id_len > 1) kc = KClamp{ch.klo, ch.khi};
        }
    }
    tot = wave_last(wave_incl_sum(meta_len(m), lane));
    // The summary of a block that updates k carries the COMPOSITION of the clamps of the segment's blocks up to
    // and including it (not its own plateau): the pack kernel then gets the block's k as one clamp of the
    // segment's carried-in k, without scanning the clamps a second time.
    const uint32_t incl_c = wave_incl_clamp(clamp_pack(kc), lane);
    cl = wave_last(incl_c);
    const uint32_t opt = meta_opt(m);
    // (clamp_pack is lo | hi << 8, the summary keeps them in its two upper bytes: one byte permute)
    if (valid && opt != OPT_ZERO && opt != OPT_ZCONT && c.id_len > 1) m = __builtin_amdgcn_perm(incl_c, m, 0x05040100u);
    return m;
}

template <int BS, int BYTES>
__device__ __forceinline__ void analyze_body(const Cfg &c, const Seg &g, uint32_t *rows, uint32_t stride,
                                             uint32_t lane, uint64_t sg, uint32_t *__restrict__ meta,
                                             uint32_t *__restrict__ seg_bits, uint16_t *__restrict__ seg_clamp,
                                             const uint32_t *direct)
{
    wave_lds_fence();
    uint32_t tot, cl;
    const uint32_t m = analyze_segment<BS, BYTES>(c, g, rows, stride, lane, tot, cl, direct);
    if (lane < g.nv) meta[g.blk0 + lane] = m;
    if (lane == 0) {
        seg_bits[sg] = tot;
        seg_clamp[sg] = (uint16_t)cl;
    }
    wave_lds_fence();
}

// ----------------------------------------------------------------------------------------------
// K2: scan of (bits, clamp) over segments -- reduce / scan partials / apply
// ----------------------------------------------------------------------------------------------
struct ScanVal {
    uint64_t bits;
    uint32_t cl;   // packed clamp
};
__device__ __forceinline__ ScanVal scan_then(ScanVal a, ScanVal b)
{
    return ScanVal{a.bits + b.bits, clamp_pack(kclamp_then(clamp_unpack(a.cl), clamp_unpack(b.cl)))};
}
__device__ __forceinline__ ScanVal scan_identity() { return ScanVal{0, clamp_pack(kclamp_identity())}; }

// ordered inclusive scan across the 256 threads of a workgroup; returns this thread's
// EXCLUSIVE prefix and the workgroup total
__device__ __forceinline__ ScanVal block_excl_scan(ScanVal v, ScanVal &total, ScanVal *sh /*[4]*/)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: segment geometry and addresses then run on the SALU)
    ScanVal inc = v;
#pragma unroll
    for (uint32_t o = 1; o < kWave; o <<= 1) {
        ScanVal t;
        t.bits = __shfl_up((unsigned long long)inc.bits, o);
        t.cl = __shfl_up(inc.cl, o);
        if (lane >= o) inc = scan_then(t, inc);
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    ScanVal wprefix = scan_identity();
    for (uint32_t w = 0; w < wave; w++) wprefix = scan_then(wprefix, sh[w]);
    total = scan_identity();
    for (uint32_t w = 0; w < (blockDim.x >> 6); w++) total = scan_then(total, sh[w]);
    ScanVal exc;
    exc.bits = __shfl_up((unsigned long long)inc.bits, 1);
    exc.cl = __shfl_up(inc.cl, 1);
    if (lane == 0) exc = scan_identity();
    __syncthreads();
    return scan_then(wprefix, exc);
}

#ifndef AEC_ENC_PART                  // (what does not depend on the block size)
constexpr uint32_t kScanItems = 8;   // kScanChunk = 256 * 8

__global__ void __launch_bounds__(256)
k_scan_reduce(const uint32_t *__restrict__ seg_bits, const uint16_t *__restrict__ seg_clamp,
              uint64_t nseg, ScanPartial *__restrict__ partials)
{
    __shared__ ScanVal sh[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanItems;
    ScanVal acc = scan_identity();
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint64_t s = base + i;
        if (s < nseg) acc = scan_then(acc, ScanVal{seg_bits[s], seg_clamp[s]});
    }
    ScanVal total;
    block_excl_scan(acc, total, sh);
    if (threadIdx.x == 0) {
        const KClamp k = clamp_unpack(total.cl);
        partials[blockIdx.x] = ScanPartial{total.bits, k.lo, k.hi};
    }
}

// single workgroup: exclusive scan of the chunk partials in place, final totals to *res
__global__ void __launch_bounds__(256)
k_scan_partials(ScanPartial *partials, uint64_t nchunks, uint32_t start_bit, uint32_t k_in,
                EncResult *res)
{
    __shared__ ScanVal sh[4];
    const uint64_t per = (nchunks + 255) / 256;
    const uint64_t lo = (uint64_t)threadIdx.x * per;
    uint64_t hi = lo + per;
    if (hi > nchunks) hi = nchunks;
    ScanVal acc = scan_identity();
    for (uint64_t i = lo; i < hi; i++)
        acc = scan_then(acc, ScanVal{partials[i].bits, clamp_pack(KClamp{partials[i].lo, partials[i].hi})});
    ScanVal total;
    ScanVal run = block_excl_scan(acc, total, sh);
    for (uint64_t i = lo; i < hi; i++) {
        const ScanVal cur{partials[i].bits, clamp_pack(KClamp{partials[i].lo, partials[i].hi})};
        const KClamp k = clamp_unpack(run.cl);
        partials[i] = ScanPartial{run.bits, k.lo, k.hi};
        run = scan_then(run, cur);
    }
    if (threadIdx.x == 0) {
        const KClamp t = clamp_unpack(total.cl);
        res->total_bits = total.bits;
        res->k_out = kclamp_apply(t, k_in);
        res->overflow = 0;
        res->k_lo = t.lo;
        res->k_hi = t.hi;
    }
    (void)start_bit;
}

__global__ void __launch_bounds__(256)
k_scan_apply(const uint32_t *__restrict__ seg_bits, const uint16_t *__restrict__ seg_clamp,
             uint64_t nseg, const ScanPartial *__restrict__ partials, uint32_t start_bit, uint32_t k_in,
             uint32_t segs_per_rsi, uint64_t rsi_count, uint64_t *__restrict__ seg_start,
             uint8_t *__restrict__ seg_kin, uint64_t *__restrict__ rsi_off, EncResult *res,
             uint32_t *__restrict__ out_words, uint64_t cap_words, uint32_t segs_per_wave,
             const ShardCarry *__restrict__ carry)
{
    __shared__ ScanVal sh[4];
    if (carry) {                       // one stream over several devices: what precedes this shard
        start_bit = (uint32_t)(carry->start_bit & 7u);
        k_in = carry->k_in;
    }
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanItems;
    ScanVal item[kScanItems];
    ScanVal acc = scan_identity();
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint64_t s = base + i;
        item[i] = (s < nseg) ? ScanVal{seg_bits[s], seg_clamp[s]} : scan_identity();
        acc = scan_then(acc, item[i]);
    }
    ScanVal total;
    ScanVal run = block_excl_scan(acc, total, sh);
    const ScanPartial cp = partials[blockIdx.x];
    run = scan_then(ScanVal{cp.bits, clamp_pack(KClamp{cp.lo, cp.hi})}, run);
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint64_t s = base + i;
        if (s < nseg) {
            const uint64_t bit = (uint64_t)start_bit + run.bits;
            seg_start[s] = bit;
            seg_kin[s] = (uint8_t)kclamp_apply(clamp_unpack(run.cl), k_in);
            if (rsi_off && (s % segs_per_rsi) == 0) rsi_off[s / segs_per_rsi] = bit;
            // k_pack writes every word of the stream with plain stores except the words two of its
            // waves share -- the one a wave's first segment starts in -- which both OR into: those
            // (and the last word of the stream, below) are all that has to be zero beforehand
            if ((s % segs_per_wave) == 0 && (bit >> 5) < cap_words) out_words[bit >> 5] = 0u;
        }
        run = scan_then(run, item[i]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        res->k_out = kclamp_apply(KClamp{res->k_lo, res->k_hi}, k_in);   // k_in is only known now
        if (rsi_off) rsi_off[rsi_count] = (uint64_t)start_bit + res->total_bits;
        const uint64_t end = (uint64_t)start_bit + res->total_bits;
        if ((end + 7) / 8 > cap_words * 4) res->overflow = 1;
        for (uint64_t w = end >> 5; w <= (end >> 5) + 1; w++)      // open last word (+ one of padding)
            if (w < cap_words) out_words[w] = 0u;
    }
}

// ---- a batch of equal, RSI-aligned chunks as ONE launch set ---------------------------------------------
// The chunks of a batch (aec_gpu_encode_uniform_batch_async) lie back to back in the input and are analysed
// and packed as one long input -- an RSI never looks across its borders -- with two differences, both in the
// scan: the k carried into a chunk is 0, and a chunk's stream starts on the byte behind the end of the one in
// front (every stream zero-padded to a byte as aec_buffer_encode pads it).  A chunk has at most kScanChunk
// segments, so one workgroup scans one chunk: totals, then the chunks' bases (one workgroup), then the start
// bit and k of every segment.
__global__ void __launch_bounds__(256)
k_batch_reduce(const uint32_t *__restrict__ seg_bits, uint32_t segs, BatchChunk *__restrict__ chunks)
{
    __shared__ ScanVal sh[4];
    const uint64_t first = (uint64_t)blockIdx.x * segs;
    ScanVal acc = scan_identity();
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint32_t s = threadIdx.x * kScanItems + i;
        if (s < segs) acc.bits += seg_bits[first + s];
    }
    ScanVal total;
    block_excl_scan(acc, total, sh);
    if (threadIdx.x == 0) chunks[blockIdx.x].bits = total.bits;
}

__global__ void __launch_bounds__(256)
k_batch_bases(BatchChunk *chunks, uint64_t n, uint64_t cap_bytes, EncResult *res)
{
    __shared__ ScanVal sh[4];
    const uint64_t per = (n + 255) / 256;
    const uint64_t lo = (uint64_t)threadIdx.x * per;
    uint64_t hi = lo + per;
    if (hi > n) hi = n;
    ScanVal acc = scan_identity();
    for (uint64_t i = lo; i < hi; i++) acc.bits += (chunks[i].bits + 7) / 8;
    ScanVal total;
    ScanVal run = block_excl_scan(acc, total, sh);
    for (uint64_t i = lo; i < hi; i++) {
        chunks[i].base_bits = run.bits * 8;
        run.bits += (chunks[i].bits + 7) / 8;
    }
    if (threadIdx.x == 0) {
        res->total_bits = total.bits * 8;
        res->k_out = 0;
        res->overflow = total.bits > cap_bytes ? 1u : 0u;
        res->k_lo = 0;
        res->k_hi = 0;
    }
}

__global__ void __launch_bounds__(256)
k_batch_apply(const uint32_t *__restrict__ seg_bits, const uint16_t *__restrict__ seg_clamp, uint32_t segs,
              const BatchChunk *__restrict__ chunks, uint64_t *__restrict__ seg_start, uint8_t *__restrict__ seg_kin,
              uint32_t *__restrict__ out_words, uint64_t cap_words, uint32_t segs_per_wave)
{
    __shared__ ScanVal sh[4];
    const uint64_t first = (uint64_t)blockIdx.x * segs;
    ScanVal item[kScanItems];
    ScanVal acc = scan_identity();
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint32_t s = threadIdx.x * kScanItems + i;
        item[i] = (s < segs) ? ScanVal{seg_bits[first + s], seg_clamp[first + s]} : scan_identity();
        acc = scan_then(acc, item[i]);
    }
    ScanVal total;
    ScanVal run = block_excl_scan(acc, total, sh);
    const uint64_t base = chunks[blockIdx.x].base_bits;
#pragma unroll
    for (uint32_t i = 0; i < kScanItems; i++) {
        const uint32_t s = threadIdx.x * kScanItems + i;
        if (s < segs) {
            const uint64_t bit = base + run.bits;
            seg_start[first + s] = bit;
            seg_kin[first + s] = (uint8_t)kclamp_apply(clamp_unpack(run.cl), 0u);
            // (as k_scan_apply: the word a wave's first segment starts in is shared with its neighbour)
            if (((first + s) % segs_per_wave) == 0 && (bit >> 5) < cap_words) out_words[bit >> 5] = 0u;
        }
        run = scan_then(run, item[i]);
    }
    if (threadIdx.x == 0) {                         // the chunk's open last word (+ one: its byte padding)
        const uint64_t end = base + total.bits;
        for (uint64_t w = end >> 5; w <= (end >> 5) + 1; w++)
            if (w < cap_words) out_words[w] = 0u;
    }
}

// segment table for segment-parallel decoding: start bit + preceding raw sample per segment
__global__ void __launch_bounds__(256)
k_seg_table(const Cfg c, const uint8_t *__restrict__ in, const uint64_t *__restrict__ seg_start,
            SegEntry *__restrict__ table)
{
    const uint64_t sg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= c.total_segs) return;
    const Seg g = seg_geom(c, sg);
    uint32_t prev = 0;
    if (g.b0 != 0) {
        uint64_t i = g.samp0 - 1;
        if (i >= c.total_samples) i = c.total_samples - 1;
        prev = load_sample_bytes(in + i * c.bytes, c.bytes, (c.flags & F_MSB) != 0);
    }
    table[sg] = SegEntry{seg_start[sg], prev, 0u};
}
#endif

// Emission of one segment into its LDS image (rows in LDS, block summaries in registers): offsets
// inside the segment by a DPP prefix sum of the lengths, k by the DPP clamp scan from the k carried
// into the segment.  Word 0 of the image continues `pending`, the open tail word of the wave's
// previous segment.  Returns the segment's bit length and clamp (wave-uniform).
template <int BS, int BYTES>
__device__ __forceinline__ void emit_segment(const Cfg &c, const Seg &g, const uint32_t *rows, uint32_t stride,
                                             uint32_t *obuf, uint32_t lane, uint32_t m, uint32_t kin, uint32_t lead,
                                             uint32_t ref_sample, uint32_t pending, uint32_t &total,
                                             const uint32_t *direct = nullptr)   // the lane's block from direct_finish
{
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const bool pp = c.flags & F_PREPROCESS;
    const bool valid = lane < g.nv;
    const uint32_t len = meta_len(m), opt = meta_opt(m);
    const uint32_t incl = wave_incl_sum(len, lane);
    total = wave_last(incl);
    const uint32_t excl = incl - len;

    // (the summary holds the composition of the segment's clamps up to and including this block, see
    // analyze_segment: the block's k is one clamp of the k carried into the segment)
    const bool updates_k = valid && opt != OPT_ZERO && opt != OPT_ZCONT && c.id_len > 1;
    const uint32_t k = updates_k ? kclamp_apply(KClamp{meta_a(m), meta_b(m)}, kin) : kin;

    // image of the segment; its first word continues the previous segment of this wave, whose
    // open tail word was kept in `pending` instead of being written out
    if (lane == 0) obuf[0] = pending;
    wave_lds_fence();
    {
        const bool emits = valid && opt != OPT_ZCONT;
        const uint32_t ref = (pp && g.b0 == 0 && lane == 0) ? 1u : 0u;
        const uint32_t karg = opt == OPT_ZERO ? meta_a(m) : k;
        BlockRegs<BS, Rows<BS, BYTES>::HALF> regs;
        const uint32_t *d;
        uint32_t dv[BS ? BS : 1];
        if (direct && BS > 0 && Rows<BS, BYTES>::HALF) {
#pragma unroll
            for (int j = 0; j < (BS ? BS / 2 : 0); j++) {
                dv[2 * j] = direct[j] & 0xFFFFu;
                dv[2 * j + 1] = direct[j] >> 16;
            }
            d = dv;
        } else {
            d = regs.load(rows + (valid ? lane : 0u) * stride);
        }
        LdsSink sink{obuf};
        BitWriter<LdsSink> bw(sink, lead + excl);
        bool done = !emits;
        if (BS > 0 && BS <= 16) {
            // small blocks: unary and field regions assembled in registers (aec_lane.h emit_small)
            uint32_t ubits = 0, fbits = 0;
            const bool small = emits && small_eligible(c, bs, opt, karg, ref, len, ubits, fbits);
            emit_small<(BS > 0 && BS <= 16 ? BS : 8)>(bw, d, c, opt, karg, ref, ref_sample, ubits, fbits, small);
            done = done || small;
        }
        if (BS == 32 || (BS == 64 && AEC_ENC_GRP64)) {
            // blocks of 32 (and 64): the split option appended in groups (aec_lane.h emit_split_groups;
            // C3 pack 2.40 -> 2.19 ms)
            const bool grp = !done && opt == OPT_SPLIT;
            emit_split_groups<(BS == 32 || BS == 64 ? BS : 32)>(bw, d, c, karg, ref, ref_sample, grp);
            done = done || grp;
        }
        if (__any(!done)) {
            if (!done) emit_block<BS>(bw, d, c, opt, karg, ref, ref_sample);
        }
    }
    wave_lds_fence();
}

// ----------------------------------------------------------------------------------------------
// K3: pack
// ----------------------------------------------------------------------------------------------
template <int BS, int BYTES>
__global__ void __launch_bounds__(256)
k_pack(const Cfg c, const uint8_t *__restrict__ in, const uint32_t *__restrict__ meta,
       const uint64_t *__restrict__ seg_start, const uint8_t *__restrict__ seg_kin,
       uint32_t *__restrict__ out_words, uint64_t cap_words, uint32_t segs_per_wave, uint32_t obuf_words,
       uint32_t fast_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: segment geometry and addresses then run on the SALU)
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const uint32_t stride = Rows<BS, BYTES>::stride_words(bs);
    const uint32_t per_wave = 64u * stride + obuf_words;
    uint32_t *rows = smem + (size_t)wave * per_wave;
    uint32_t *obuf = rows + 64u * stride;
    const bool pp = c.flags & F_PREPROCESS, msb = c.flags & F_MSB;

    const uint64_t gwave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    uint64_t sg = gwave * segs_per_wave;
    uint64_t sg_end = sg + segs_per_wave;
    if (sg_end > c.total_segs) sg_end = c.total_segs;

    Feeder<BS, BYTES> feeder;
    feeder.init(c, fast_ok);
    Seg gnext = seg_geom(c, sg < sg_end ? sg : 0);
    if (sg < sg_end) {
        if (Feeder<BS, BYTES>::DIRECT) feeder.prefetch_direct(c, in, gnext, lane);
        else feeder.prefetch(c, in, gnext, lane);
    }
    uint32_t pending = 0;        // open tail word of the previous segment (stream bit order)
    bool first_seg = true, carried_shared = false;
    // the image buffer starts out zero and every word is zeroed again when it is copied out
    for (uint32_t w = lane; w < obuf_words; w += kWave) obuf[w] = 0u;

    // what a segment needs from HBM besides its samples
    struct SegIn {
        uint32_t m, kin, ref_sample;
        uint64_t start;
    };
    auto seg_in = [&](const Seg &g, uint64_t sgi) {
        SegIn r;
        r.m = lane < g.nv ? meta[g.blk0 + lane] : meta_pack(0, OPT_ZCONT, 0, 0);
        r.kin = seg_kin[sgi];
        r.start = seg_start[sgi];
        r.ref_sample = 0;
        if (pp && g.b0 == 0 && lane == 0)
            r.ref_sample = load_sample_bytes(in + g.samp0 * c.bytes, c.bytes, msb) & low_mask32(c.bps);
        return r;
    };
    // emission of one segment from its rows and copy-out of the image
    auto do_segment = [&](const Seg &g, const uint32_t *seg_rows, const SegIn &si, uint64_t sgi, const uint32_t *direct) {
        const uint32_t lead = (uint32_t)(si.start & 31u);
        uint32_t total;
        emit_segment<BS, BYTES>(c, g, seg_rows, stride, obuf, lane, si.m, si.kin, lead, si.ref_sample, pending, total, direct);
        const uint32_t nwords = (lead + total + 31u) >> 5;

        // Copy the image out.  Only a word this wave does not own alone needs an atomic: the first
        // word of the wave's first segment (shared with the previous wave) and the open tail word of
        // its last segment.  An open tail in between is carried to the next segment in `pending`.
        const uint64_t gw = si.start >> 5;
        const uint32_t tail = (lead + total) & 31u;
        const bool last_seg = sgi + 1 == sg_end;
        const bool carry_tail = tail != 0 && !last_seg && nwords > 0;
        // word 0 also holds bits of another wave only in the wave's first segment, or when a one-word
        // segment carried that word along
        const bool left_shared = first_seg ? lead != 0 : carried_shared;
        const uint32_t tail_word = carry_tail ? obuf[nwords - 1] : 0u;   // uniform: every lane reads the same word
        for (uint32_t w = lane; w < nwords; w += kWave) {
            const uint32_t v = obuf[w];
            obuf[w] = 0u;
            const uint64_t idx = gw + w;
            const bool is_tail = w == nwords - 1 && tail != 0;
            if (idx < cap_words && !(is_tail && carry_tail)) {
                const bool shared = (w == 0 && left_shared) || is_tail;
                const uint32_t sv = bswap32(v);
                if (!shared)
                    out_words[idx] = sv;                 // (zero words too: nothing clears the buffer)
                else if (v != 0)
                    __hip_atomic_fetch_or(&out_words[idx], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        pending = tail_word;
        carried_shared = carry_tail && nwords == 1 && left_shared;
        first_seg = false;
        wave_lds_fence();
    };

    if (Feeder<BS, BYTES>::DIRECT) {
        // small blocks: lane = block from the load on, nothing goes through the rows (Feeder::DIRECT)
        for (; sg < sg_end; sg++) {
            const auto cur = feeder.pre_direct;
            const Seg g = gnext;
            const SegIn si = seg_in(g, sg);
            if (sg + 1 < sg_end) gnext = seg_next(c, gnext);
            feeder.prefetch_direct(c, in, gnext, lane);   // the next segment's loads fly during this one
            if (feeder.direct_ok(c, g)) {
                uint32_t w[BS ? BS / 2 : 1];
                direct_finish<(Feeder<BS, BYTES>::DIRECT ? BS : 8), (Feeder<BS, BYTES>::DIRECT ? BYTES : 1)>(c, g, cur, lane, w);
                do_segment(g, rows, si, sg, w);
            } else {
                feeder.feed_now(c, in, g, rows, stride, lane);
                do_segment(g, rows, si, sg, nullptr);
            }
        }
        return;
    }
    for (; sg < sg_end; sg++) {
        const auto cur = feeder.pre;
        const Seg g = gnext;
        // everything this segment needs from HBM is requested before the first wait (requesting the
        // summaries a segment ahead as well was tried: no gain, two registers too many)
        const SegIn si = seg_in(g, sg);
        if (sg + 1 < sg_end) gnext = seg_next(c, gnext);
        feeder.prefetch(c, in, gnext, lane);      // next segment's loads fly during this one
        feeder.feed(c, in, g, cur, rows, stride, lane);
        do_segment(g, rows, si, sg, nullptr);
    }
}

// ----------------------------------------------------------------------------------------------
// K1+K2+K3 in ONE pass: analysis, scan and emission without a second read of the input
// ----------------------------------------------------------------------------------------------
// A workgroup takes a ticket (its partition = the next waves x SEGS segments, so every partition it
// will ever wait for has started before it), analyses its segments ONCE -- the preprocessed rows of
// all of them stay in LDS, the block summaries in registers -- and publishes the partition's
// (bit length, k clamp).  Start bit and carried k come from a decoupled look-back over those
// aggregates (Merrill & Garland's single-pass scan: the nearest predecessor that already knows its
// inclusive prefix, plus the aggregates after it), then every wave emits from the rows in LDS.
// Nothing but the input is read from HBM and nothing but the stream is written: no per-block
// summaries, no scan kernels, no second pass over the input (reference src/encode.c:235-311,
// 313-434, 520-583 stay bit-exact: same arithmetic, other order).
//
// Hand-offs between workgroups follow /opt/skills/guides/cdna_hip_programming.md Guideline 16, form
// R2: every shared word is ONE naturally aligned 8-byte granule {status, value} written by a relaxed
// agent-scope atomic store and polled by relaxed agent-scope atomic loads (the per-XCD L2s are not
// coherent); all of them are zeroed by a hipMemsetAsync in front of the launch.  Every spin is
// bounded: a timeout sets *fail (the host reports AEC_MEM_ERROR) instead of hanging the device.
//   desc[p]   [0:2) status (1 = the partition's own aggregate, 2 = inclusive prefix), [2:7) clamp lo,
//             [7:12) clamp hi, [12:64) bits
//   tails[p]  bit 63 valid, bit 62 "the partition ends inside a word", [0:32) that open word (the
//             stream words two partitions share are written by the later one)
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ uint64_t granule_load(const uint64_t *p)
{
    return __hip_atomic_load(reinterpret_cast<gu64 *>(reinterpret_cast<uintptr_t>(p)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(reinterpret_cast<gu64 *>(reinterpret_cast<uintptr_t>(p)), v, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t desc_pack(uint32_t status, ScanVal v)
{
    return (uint64_t)status | ((uint64_t)(v.cl & 0x1Fu) << 2) | ((uint64_t)((v.cl >> 8) & 0x1Fu) << 7) | (v.bits << 12);
}
__device__ __forceinline__ ScanVal desc_unpack(uint64_t x)
{
    return ScanVal{x >> 12, (uint32_t)((x >> 2) & 0x1Fu) | ((uint32_t)((x >> 7) & 0x1Fu) << 8)};
}

// ordered reduction over the 64 lanes (lane 0 first); result valid in every lane
__device__ __forceinline__ ScanVal wave_fold(ScanVal v, uint32_t lane)
{
#pragma unroll
    for (uint32_t o = 1; o < kWave; o <<= 1) {
        ScanVal t;
        t.bits = __shfl_up((unsigned long long)v.bits, o);
        t.cl = __shfl_up(v.cl, o);
        if (lane >= o) v = scan_then(t, v);
    }
    ScanVal r;
    r.bits = __shfl((unsigned long long)v.bits, 63);
    r.cl = __shfl(v.cl, 63);
    return r;
}

constexpr uint32_t kSpinLimit = 1u << 22;      // x (s_sleep + an L2 round trip): seconds, never reached in a healthy run

// exclusive prefix of partition p (whole wave; every lane returns it)
__device__ __forceinline__ ScanVal lookback(const uint64_t *desc, uint32_t p, uint32_t lane, uint32_t *fail)
{
    ScanVal acc = scan_identity();               // composition of the partitions after `idx`, up to p - 1
    int64_t idx = (int64_t)p - 1;
    for (;;) {
        // ascending lane = ascending partition: lane j looks at partition idx - (63 - j); in front of the
        // stream stands a virtual partition that "knows" the empty prefix
        const int64_t q = idx - (int64_t)(63u - lane);
        uint64_t x = 2u | ((uint64_t)31u << 7);
        for (uint32_t spins = 0;; spins++) {
            if (q >= 0) x = granule_load(desc + q);
            if (__all((x & 3u) != 0u)) break;
            if (spins > kSpinLimit) {
                if (lane == 0) atomicOr(fail, 1u);
                return acc;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const uint64_t incl = __ballot((x & 3u) == 2u);
        ScanVal v = desc_unpack(x);
        if (incl) {
            const uint32_t jl = 63u - (uint32_t)__builtin_clzll(incl);   // nearest one with an inclusive prefix
            if (lane < jl) v = scan_identity();
            return scan_then(wave_fold(v, lane), acc);
        }
        acc = scan_then(wave_fold(v, lane), acc);
        idx -= 64;
    }
}

struct WaveEdge {          // the words a wave does not store itself
    uint64_t head_idx, tail_idx;
    uint32_t head_val, tail_val;
    uint32_t shared_head, tail_open;
};

template <int BS, int BYTES, int SEGS>
__global__ void __launch_bounds__(256)
k_encode_fused(const Cfg c, const uint8_t *__restrict__ in, uint32_t *__restrict__ out_words, uint64_t cap_words,
               uint64_t *desc, uint64_t *tails, uint32_t *ticket, uint32_t *fail, uint32_t nparts,
               uint32_t parts_per_wg, uint32_t start_bit, uint32_t k_in, uint64_t *__restrict__ rsi_off,
               SegEntry *__restrict__ seg_table, EncResult *res, uint32_t obuf_words, uint32_t fast_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t sh_part;
    __shared__ ScanVal sh_agg[4], sh_excl;
    __shared__ WaveEdge sh_edge[4];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const uint32_t bs = BS ? (uint32_t)BS : c.bs;
    const uint32_t stride = Rows<BS, BYTES>::stride_words(bs);
    const uint32_t seg_words = 64u * stride;
    uint32_t *rows0 = smem + (size_t)wave * (SEGS * seg_words + obuf_words);
    uint32_t *obuf = rows0 + SEGS * seg_words;
    const bool pp = c.flags & F_PREPROCESS, msb = c.flags & F_MSB;

    // One ticket buys parts_per_wg consecutive partitions.  Whatever a partition waits for belongs to a
    // lower ticket -- a workgroup that is resident and only ever waits for still lower ones -- or to this
    // workgroup's own earlier rounds.  (More than one partition per ticket SERIALISES the grid: the first
    // partition of ticket T needs the aggregate of the last partition of ticket T-1, which that workgroup
    // only reaches after emitting its earlier ones -- measured 2.5 s instead of 3 ms at 4 GiB.  Kept as a
    // knob for experiments; the default is 1.)
    if (threadIdx.x == 0) sh_part = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t p_first = sh_part * parts_per_wg;
    for (uint32_t w = lane; w < obuf_words; w += kWave) obuf[w] = 0u;   // the image buffer starts out zero
    for (uint32_t p = p_first; p < p_first + parts_per_wg && p < nparts; p++) {
    const uint64_t sg0 = ((uint64_t)p * nwaves + wave) * SEGS;
    const uint32_t nseg = sg0 >= c.total_segs ? 0u : (c.total_segs - sg0 < (uint64_t)SEGS ? (uint32_t)(c.total_segs - sg0) : (uint32_t)SEGS);

    // ---- phase 1: every segment of the wave analysed once; rows stay in LDS, summaries in registers
    Feeder<BS, BYTES> feeder;
    feeder.init(c, fast_ok);
    uint32_t m[SEGS], seg_cl[SEGS];
    ScanVal wagg = scan_identity();
    const Seg g0 = seg_geom(c, nseg ? sg0 : 0);
    {
        Seg g = g0;
        if (nseg) feeder.prefetch(c, in, g, lane);
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            m[s] = meta_pack(0, OPT_ZCONT, 0, 0);
            seg_cl[s] = clamp_pack(kclamp_identity());
            if ((uint32_t)s < nseg) {
                const auto cur = feeder.pre;
                const Seg gcur = g;
                if ((uint32_t)s + 1 < nseg) {
                    g = seg_next(c, g);
                    feeder.prefetch(c, in, g, lane);      // next segment's loads fly during this one
                }
                uint32_t *rows = rows0 + s * seg_words;
                feeder.feed(c, in, gcur, cur, rows, stride, lane);
                wave_lds_fence();
                uint32_t tot, cl;
                m[s] = analyze_segment<BS, BYTES>(c, gcur, rows, stride, lane, tot, cl);
                seg_cl[s] = cl;
                wagg = scan_then(wagg, ScanVal{tot, cl});
            }
        }
    }
    if (lane == 0) sh_agg[wave] = wagg;
    __syncthreads();

    // ---- the partition's place in the stream
    if (wave == 0) {
        ScanVal agg = scan_identity();
        for (uint32_t w = 0; w < nwaves; w++) agg = scan_then(agg, sh_agg[w]);
        if (lane == 0) granule_store(desc + p, desc_pack(1u, agg));
        const ScanVal excl = lookback(desc, p, lane, fail);
        const ScanVal inc = scan_then(excl, agg);
        if (lane == 0) {
            granule_store(desc + p, desc_pack(2u, inc));
            sh_excl = excl;
            if (p == nparts - 1u) {                      // the stream's totals
                const KClamp t = clamp_unpack(inc.cl);
                const uint64_t end = (uint64_t)start_bit + inc.bits;
                res->total_bits = inc.bits;
                res->k_out = kclamp_apply(t, k_in);
                res->overflow = (end + 7) / 8 > cap_words * 4 ? 1u : 0u;
                res->k_lo = t.lo;
                res->k_hi = t.hi;
                if (rsi_off) rsi_off[c.rsi_count] = end;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: emission from the rows in LDS
    ScanVal run = sh_excl;
    for (uint32_t w = 0; w < wave; w++) run = scan_then(run, sh_agg[w]);
    const uint64_t wave_start = (uint64_t)start_bit + run.bits;
    const uint64_t head_idx = wave_start >> 5;
    const bool shared_head = (wave_start & 31u) != 0;
    uint32_t head_val = 0, pending = 0;
    uint64_t pos = wave_start;
    {
        Seg g = g0;
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            if ((uint32_t)s < nseg) {
                const Seg gcur = g;
                if ((uint32_t)s + 1 < nseg) g = seg_next(c, g);
                const uint32_t kin = kclamp_apply(clamp_unpack(run.cl), k_in);
                const uint64_t start = pos;
                uint32_t ref_sample = 0;
                if (lane == 0) {
                    if (pp && gcur.b0 == 0)
                        ref_sample = load_sample_bytes(in + gcur.samp0 * c.bytes, c.bytes, msb) & low_mask32(c.bps);
                    if (rsi_off && gcur.b0 == 0) rsi_off[gcur.rsi_idx] = start;
                    if (seg_table) {
                        uint32_t prev = 0;
                        if (gcur.b0 != 0) {
                            uint64_t i = gcur.samp0 - 1;
                            if (i >= c.total_samples) i = c.total_samples - 1;
                            prev = load_sample_bytes(in + i * c.bytes, c.bytes, msb);
                        }
                        seg_table[sg0 + s] = SegEntry{start, prev, 0u};
                    }
                }
                const uint32_t lead = (uint32_t)(start & 31u);
                uint32_t total;
                emit_segment<BS, BYTES>(c, gcur, rows0 + s * seg_words, stride, obuf, lane, m[s], kin, lead, ref_sample,
                                        pending, total);
                const uint32_t nwords = (lead + total + 31u) >> 5;
                const uint64_t gw = start >> 5;
                const bool open_tail = ((lead + total) & 31u) != 0;
                // The open tail word goes on into the wave's next segment (or, after the last one, to
                // whoever writes the word the wave ends in); the word the wave STARTS in, when it also
                // holds bits of the previous wave, is left to the boundary pass below.
                const uint32_t tail_word = open_tail ? obuf[nwords - 1] : 0u;
                if (gw == head_idx) head_val = obuf[0];
                for (uint32_t w = lane; w < nwords; w += kWave) {
                    const uint32_t v = obuf[w];
                    obuf[w] = 0u;
                    const uint64_t idx = gw + w;
                    const bool is_tail = w == nwords - 1 && open_tail;
                    const bool is_head = idx == head_idx && shared_head;
                    if (idx < cap_words && !is_tail && !is_head) out_words[idx] = bswap32(v);
                }
                pending = tail_word;
                pos += total;
                run = scan_then(run, ScanVal{total, seg_cl[s]});
                wave_lds_fence();
            }
        }
    }
    if (lane == 0) {
        WaveEdge e;
        e.head_idx = head_idx;
        e.head_val = head_val;
        e.shared_head = shared_head ? 1u : 0u;
        e.tail_idx = pos >> 5;
        e.tail_open = (pos & 31u) != 0 ? 1u : 0u;
        e.tail_val = pending;
        sh_edge[wave] = e;
    }
    __syncthreads();

    // ---- the words waves / partitions share: one lane walks the (at most four) wave edges in order
    if (threadIdx.x == 0) {
        uint32_t nact = 0;
        for (uint32_t w = 0; w < nwaves; w++)
            if (((uint64_t)p * nwaves + w) * SEGS < c.total_segs) nact = w + 1;
        const uint64_t part_start = (uint64_t)start_bit + sh_excl.bits;
        const bool lead_open = (part_start & 31u) != 0;
        bool all_pass = true;                               // every wave begins and ends inside ONE word
        for (uint32_t w = 0; w < nact; w++) {
            const WaveEdge &e = sh_edge[w];
            all_pass = all_pass && e.shared_head && e.tail_open && e.tail_idx == e.head_idx;
        }
        uint32_t c_open = 0, c_val = 0;
        uint64_t c_idx = 0;
        auto resolve = [&](uint32_t incoming, bool store) {
            c_open = lead_open ? 1u : 0u;
            c_val = incoming;
            c_idx = part_start >> 5;
            for (uint32_t w = 0; w < nact; w++) {
                const WaveEdge &e = sh_edge[w];
                if (e.shared_head) {
                    const uint32_t val = e.head_val | (c_open ? c_val : 0u);
                    if (e.tail_open && e.tail_idx == e.head_idx) {      // still the same word: pass it on
                        c_open = 1u;
                        c_val = val;
                        c_idx = e.head_idx;
                        continue;
                    }
                    if (store && e.head_idx < cap_words) out_words[e.head_idx] = bswap32(val);
                }
                c_open = e.tail_open;
                c_val = e.tail_val;
                c_idx = e.tail_idx;
            }
        };
        auto publish = [&]() {
            granule_store(tails + p, (1ull << 63) | ((uint64_t)c_open << 62) | c_val);
        };
        // what leaves this partition does not depend on what enters it unless every wave passes the
        // word on: publish first, so that the chain of partitions never waits on more than one hop
        if (!all_pass) {
            resolve(0u, false);
            publish();
        }
        uint32_t incoming = 0;
        if (lead_open && p > 0) {
            uint64_t x = 0;
            for (uint32_t spins = 0;; spins++) {
                x = granule_load(tails + (p - 1));
                if (x >> 63) break;
                if (spins > kSpinLimit) {
                    atomicOr(fail, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            incoming = (uint32_t)x;
        }
        resolve(incoming, true);
        if (all_pass) publish();
        if (p == nparts - 1u) {                              // the stream's last word, zero padded, and one more
            uint64_t z = c_idx;
            if (c_open) {
                if (c_idx < cap_words) out_words[c_idx] = bswap32(c_val);
                z = c_idx + 1;
            }
            for (uint64_t w = z; w <= c_idx + 1; w++)
                if (w < cap_words) out_words[w] = 0u;
        }
    }
    __syncthreads();       // the shared records are reused by the next partition
    }
}

// a look-back that timed out (never in a healthy run) must not pass for a result
__global__ void k_fused_finish(const uint32_t *fail, EncResult *res)
{
    if (*fail) res->overflow = 2u;
}

// ----------------------------------------------------------------------------------------------
// dispatch
// ----------------------------------------------------------------------------------------------
struct LaunchGeom {
    uint32_t waves_per_block;
    uint32_t segs_per_wave;
    uint32_t grid;
    size_t lds_bytes;
    uint32_t obuf_words;
};

LaunchGeom make_geom(const Cfg &c, bool with_obuf)
{
    LaunchGeom g;
    const bool templated = c.bs == 8 || c.bs == 16 || c.bs == 32 || c.bs == 64;
    const uint32_t stride = (templated && c.bytes <= 2) ? c.bs / 2 + 4 : c.bs + 4;   // Rows<>::stride_words
    const uint32_t maxlen = c.id_len + c.bs * c.bps + 2 + c.bps;
    g.obuf_words = with_obuf ? ((64u * maxlen + 62u) / 32u + 4u) & ~3u : 0u;
    const size_t per_wave = ((size_t)64 * stride + g.obuf_words) * 4;
    uint32_t wpb = (uint32_t)(65536 / per_wave);
    if (wpb > 4) wpb = 4;
    if (wpb < 1) wpb = 1;
    g.waves_per_block = wpb;
    g.lds_bytes = per_wave * wpb;
    // A wave walks a run of consecutive segments: interior word boundaries then stay inside the
    // wave (no atomics) and the next segment's loads overlap the current one.  Small inputs are
    // spread over the chip instead.
    uint32_t spw = 8;
    while (spw > 1 && c.total_segs < (uint64_t)spw * wpb * 2048) spw >>= 1;
    g.segs_per_wave = spw;
    const uint64_t waves = (c.total_segs + spw - 1) / spw;
    g.grid = (uint32_t)((waves + wpb - 1) / wpb);
    return g;
}

template <int BS, int BYTES>
void launch_analyze_t(const Cfg &c, const uint8_t *in, const EncWorkspace &ws, uint32_t fast_ok,
                      hipStream_t st)
{
    const LaunchGeom g = make_geom(c, false);
    hipLaunchKernelGGL((k_analyze<BS, BYTES>), dim3(g.grid), dim3(64 * g.waves_per_block), g.lds_bytes, st,
                       c, in, ws.meta, ws.seg_bits, ws.seg_clamp, g.segs_per_wave, fast_ok);
}

template <int BS, int BYTES>
void launch_pack_t(const Cfg &c, const uint8_t *in, const EncWorkspace &ws, uint32_t *out_words,
                   uint64_t cap_words, uint32_t fast_ok, hipStream_t st)
{
    const LaunchGeom g = make_geom(c, true);
    hipLaunchKernelGGL((k_pack<BS, BYTES>), dim3(g.grid), dim3(64 * g.waves_per_block), g.lds_bytes, st,
                       c, in, ws.meta, ws.seg_start, ws.seg_kin, out_words, cap_words, g.segs_per_wave,
                       g.obuf_words, fast_ok);
}

template <int BS>
void dispatch_bytes(bool pack, const Cfg &c, const uint8_t *in, const EncWorkspace &ws,
                    uint32_t *out_words, uint64_t cap_words, uint32_t fast_ok, hipStream_t st)
{
#define AEC_GO(B)                                                                     \
    do {                                                                              \
        if (pack) launch_pack_t<BS, B>(c, in, ws, out_words, cap_words, fast_ok, st); \
        else launch_analyze_t<BS, B>(c, in, ws, fast_ok, st);                         \
    } while (0)
    switch (c.bytes) {
    case 1: AEC_GO(1); break;
    case 2: AEC_GO(2); break;
    case 3: AEC_GO(3); break;
    default: AEC_GO(4); break;
    }
#undef AEC_GO
}

// waves per workgroup and segments per wave of the fused kernel: as many segments per wave as keep
// three or more workgroups on a CU (the look-back costs per partition), at least one
FusedGeom fused_geom(const Cfg &c)
{
    FusedGeom g;
    const uint32_t stride = (c.bytes <= 2) ? c.bs / 2 + 4 : c.bs + 4;          // Rows<>::stride_words
    const uint32_t maxlen = c.id_len + c.bs * c.bps + 2 + c.bps;
    g.obuf_words = ((64u * maxlen + 62u) / 32u + 4u) & ~3u;
    const size_t seg_bytes = (size_t)64 * stride * 4, obuf_bytes = (size_t)g.obuf_words * 4;
    const uint32_t force = tune("AEC_FUSED_SEGS", 0);
    g.waves = 4;
    g.segs = 1;
    for (uint32_t s : {4u, 2u, 1u}) {
        if ((seg_bytes * s + obuf_bytes) * 4 <= 53 * 1024 || s == 1) {
            g.segs = s;
            break;
        }
    }
    if (force == 1 || force == 2 || force == 4) g.segs = force;
    while (g.waves > 1 && (seg_bytes * g.segs + obuf_bytes) * g.waves > 156 * 1024) g.waves >>= 1;
    g.lds_bytes = (seg_bytes * g.segs + obuf_bytes) * g.waves;
    const uint64_t per = (uint64_t)g.waves * g.segs;
    g.nparts = (uint32_t)((c.total_segs + per - 1) / per);
    g.parts_per_wg = 1u;
    const uint32_t ppw = tune("AEC_FUSED_PARTS", 0);
    if (ppw >= 1 && ppw <= 64) g.parts_per_wg = ppw;
    g.grid = (g.nparts + g.parts_per_wg - 1) / g.parts_per_wg;
    return g;
}

template <int BS, int BYTES, int SEGS>
void launch_fused_t(const Cfg &c, const uint8_t *in, uint32_t *out_words, uint64_t cap_words, const FusedGeom &g,
                    void *ctl, uint32_t start_bit, uint32_t k_in, uint64_t *rsi_off, SegEntry *seg_table,
                    EncResult *res, uint32_t fast_ok, hipStream_t st)
{
    static bool big_lds = false;          // (per instantiation; benign if two threads set it twice)
    if (!big_lds) {
        // (the kernel also has a few hundred bytes of static LDS: ask for less than the CU's 160 KiB)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_encode_fused<BS, BYTES, SEGS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
            (void)hipGetLastError();
        big_lds = true;
    }
    uint8_t *b = static_cast<uint8_t *>(ctl);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(b);
    uint32_t *fail = ticket + 1;
    uint64_t *desc = reinterpret_cast<uint64_t *>(b + 16);
    uint64_t *tails = desc + g.nparts;
    hipLaunchKernelGGL((k_encode_fused<BS, BYTES, SEGS>), dim3(g.grid), dim3(64 * g.waves), g.lds_bytes, st, c, in,
                       out_words, cap_words, desc, tails, ticket, fail, g.nparts, g.parts_per_wg, start_bit, k_in,
                       rsi_off, seg_table, res, g.obuf_words, fast_ok);
    hipLaunchKernelGGL(k_fused_finish, dim3(1), dim3(1), 0, st, fail, res);
}

template <int BS, int BYTES>
void launch_fused_segs(const Cfg &c, const uint8_t *in, uint32_t *out_words, uint64_t cap_words, const FusedGeom &g,
                       void *ctl, uint32_t start_bit, uint32_t k_in, uint64_t *rsi_off, SegEntry *seg_table,
                       EncResult *res, uint32_t fast_ok, hipStream_t st)
{
    switch (g.segs) {
    case 4: launch_fused_t<BS, BYTES, 4>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    case 2: launch_fused_t<BS, BYTES, 2>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    default: launch_fused_t<BS, BYTES, 1>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    }
}

template <int BS>
void launch_fused_bytes(const Cfg &c, const uint8_t *in, uint32_t *out_words, uint64_t cap_words, const FusedGeom &g,
                        void *ctl, uint32_t start_bit, uint32_t k_in, uint64_t *rsi_off, SegEntry *seg_table,
                        EncResult *res, uint32_t fast_ok, hipStream_t st)
{
    switch (c.bytes) {
    case 1: launch_fused_segs<BS, 1>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    case 2: launch_fused_segs<BS, 2>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    case 3: launch_fused_segs<BS, 3>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    default: launch_fused_segs<BS, 4>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st); break;
    }
}

}  // namespace

#ifdef AEC_ENC_PART
// ---- the kernels of ONE block size (this object: -DAEC_ENC_PART=<block size>, 0 = any other) --------------------------------
template <int BS>
void enc_part(bool pack, const Cfg &c, const uint8_t *in, const EncWorkspace &ws, uint32_t *out_words, uint64_t cap_words,
              uint32_t fast_ok, hipStream_t st)
{
    if constexpr (BS == 0) {
        if (pack) launch_pack_t<0, 0>(c, in, ws, out_words, cap_words, 0, st);
        else launch_analyze_t<0, 0>(c, in, ws, 0, st);
    } else {
        dispatch_bytes<BS>(pack, c, in, ws, out_words, cap_words, fast_ok, st);
    }
}
template <int BS>
void enc_part_fused(const Cfg &c, const uint8_t *in, uint32_t *out_words, uint64_t cap_words, const FusedGeom &g, void *ctl,
                    uint32_t start_bit, uint32_t k_in, uint64_t *rsi_off, SegEntry *seg_table, EncResult *res,
                    uint32_t fast_ok, hipStream_t st)
{
    launch_fused_bytes<BS>(c, in, out_words, cap_words, g, ctl, start_bit, k_in, rsi_off, seg_table, res, fast_ok, st);
}
template void enc_part<AEC_ENC_PART>(bool, const Cfg &, const uint8_t *, const EncWorkspace &, uint32_t *, uint64_t, uint32_t,
                                     hipStream_t);
#if AEC_ENC_PART != 0
template void enc_part_fused<AEC_ENC_PART>(const Cfg &, const uint8_t *, uint32_t *, uint64_t, const FusedGeom &, void *,
                                           uint32_t, uint32_t, uint64_t *, SegEntry *, EncResult *, uint32_t, hipStream_t);
#endif

#else       // ---- the object without a part ------------------------------------------------------------------------
#define AEC_ENC_EXTERN(BS)                                                                                                  \
    extern template void enc_part<BS>(bool, const Cfg &, const uint8_t *, const EncWorkspace &, uint32_t *, uint64_t, uint32_t, \
                                      hipStream_t);
AEC_ENC_EXTERN(0) AEC_ENC_EXTERN(8) AEC_ENC_EXTERN(16) AEC_ENC_EXTERN(32) AEC_ENC_EXTERN(64)
#undef AEC_ENC_EXTERN
#define AEC_ENC_EXTERN(BS)                                                                                                  \
    extern template void enc_part_fused<BS>(const Cfg &, const uint8_t *, uint32_t *, uint64_t, const FusedGeom &, void *,    \
                                            uint32_t, uint32_t, uint64_t *, SegEntry *, EncResult *, uint32_t, hipStream_t);
AEC_ENC_EXTERN(8) AEC_ENC_EXTERN(16) AEC_ENC_EXTERN(32) AEC_ENC_EXTERN(64)
#undef AEC_ENC_EXTERN

static void dispatch(bool pack, const Cfg &c, const uint8_t *in, const EncWorkspace &ws, uint32_t *out_words,
                     uint64_t cap_words, uint32_t fast_ok, hipStream_t st)
{
    switch (c.bs) {
    case 8: enc_part<8>(pack, c, in, ws, out_words, cap_words, fast_ok, st); break;
    case 16: enc_part<16>(pack, c, in, ws, out_words, cap_words, fast_ok, st); break;
    case 32: enc_part<32>(pack, c, in, ws, out_words, cap_words, fast_ok, st); break;
    case 64: enc_part<64>(pack, c, in, ws, out_words, cap_words, fast_ok, st); break;
    default: enc_part<0>(pack, c, in, ws, out_words, cap_words, fast_ok, st); break;
    }
}

// The single-pass kernel is bit-exact (tests/fused_edges.py, and the whole 4 GiB stream of the bench
// against the CPU reference) but SLOWER than the analyze / scan / pack trio on MI355X: 5.5 ms against
// 3.74 ms at C2 (4 GiB; 6.9 ms with 4 segments per wave, 7.4 ms with 1).  Keeping the rows of a wave's
// segments in LDS allows 2-4 workgroups per CU, and the serial sections of a partition -- the ticket,
// the look-back done by one wave while the others wait at the barrier, the boundary words -- take
// about 7 of the ~15 microseconds a partition lives; nothing is there to hide them behind.  So the
// two-pass kernels stay the default and AEC_ENC_FUSED=1 selects this one (kept under test).
bool fused_supported(const Cfg &c)
{
    static const bool on = tune_set("AEC_ENC_FUSED");
    return on && (c.bs == 8 || c.bs == 16 || c.bs == 32 || c.bs == 64) && c.total_segs != 0;
}

size_t fused_ctl_bytes(const Cfg &c)
{
    const FusedGeom g = fused_geom(c);
    return ((size_t)16 + (size_t)g.nparts * 16 + 15) & ~(size_t)15;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t enc_workspace_bytes(const Cfg &c, size_t *off_meta, size_t *off_bits, size_t *off_clamp,
                           size_t *off_start, size_t *off_kin, size_t *off_part)
{
    size_t o = 0;
    const uint64_t nseg = c.total_segs ? c.total_segs : 1;
    const uint64_t nchunks = (nseg + kScanChunk - 1) / kScanChunk + 1;
    *off_meta = o;  o = align_up(o + (c.total_blocks + 1) * 4, 256);
    *off_bits = o;  o = align_up(o + nseg * 4, 256);
    *off_clamp = o; o = align_up(o + nseg * 2, 256);
    *off_start = o; o = align_up(o + nseg * 8, 256);
    *off_kin = o;   o = align_up(o + nseg, 256);
    *off_part = o;  o = align_up(o + nchunks * sizeof(ScanPartial), 256);
    return o;
}

void launch_encode(const Cfg &c, const uint8_t *d_in, uint8_t *d_out, size_t out_cap,
                   uint32_t start_bit, uint32_t k_in, const EncWorkspace &ws, uint64_t *d_rsi_off,
                   EncResult *d_res, hipStream_t st, const PhaseEvents *prof, uint32_t phases,
                   SegEntry *d_seg_table, const ShardCarry *d_carry)
{
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(prof->ev[i], st); };
    uint32_t *out_words = reinterpret_cast<uint32_t *>(d_out);
    const uint64_t cap_words = out_cap / 4;
    const uint64_t nseg = c.total_segs;
    const uint64_t nchunks = (nseg + kScanChunk - 1) / kScanChunk;
    // fast loads need every segment to start on a 16-byte boundary of a 16-byte aligned buffer
    const uint32_t fast_ok = ((reinterpret_cast<uintptr_t>(d_in) & 15u) == 0 &&
                              ((uint64_t)c.rsi * c.bs * c.bytes) % 16 == 0) ? 1u : 0u;

    if (phases == ENC_ALL && ws.fused_ctl && !d_carry && fused_supported(c)) {
        // single pass: ticket, fail flag and the look-back granules are zeroed, then ONE kernel
        const FusedGeom g = fused_geom(c);
        mark(0);
        mark(1);
        (void)hipMemsetAsync(ws.fused_ctl, 0, fused_ctl_bytes(c), st);
        mark(2);
        mark(3);
        switch (c.bs) {
        case 8: enc_part_fused<8>(c, d_in, out_words, cap_words, g, ws.fused_ctl, start_bit, k_in, d_rsi_off, d_seg_table, d_res, fast_ok, st); break;
        case 16: enc_part_fused<16>(c, d_in, out_words, cap_words, g, ws.fused_ctl, start_bit, k_in, d_rsi_off, d_seg_table, d_res, fast_ok, st); break;
        case 32: enc_part_fused<32>(c, d_in, out_words, cap_words, g, ws.fused_ctl, start_bit, k_in, d_rsi_off, d_seg_table, d_res, fast_ok, st); break;
        default: enc_part_fused<64>(c, d_in, out_words, cap_words, g, ws.fused_ctl, start_bit, k_in, d_rsi_off, d_seg_table, d_res, fast_ok, st); break;
        }
        mark(4);
        return;
    }
    if (phases & ENC_PLAN) {
        mark(0);
        if (nseg) dispatch(false, c, d_in, ws, nullptr, 0, fast_ok, st);
        mark(1);
        if (nchunks)
            hipLaunchKernelGGL(k_scan_reduce, dim3((uint32_t)nchunks), dim3(256), 0, st, ws.seg_bits,
                               ws.seg_clamp, nseg, ws.partials);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, st, ws.partials, nchunks, start_bit, k_in,
                           d_res);
    }
    if (phases & ENC_EMIT) {
        if (nchunks)
            hipLaunchKernelGGL(k_scan_apply, dim3((uint32_t)nchunks), dim3(256), 0, st, ws.seg_bits,
                               ws.seg_clamp, nseg, ws.partials, start_bit, k_in, c.segs_per_rsi, c.rsi_count,
                               ws.seg_start, ws.seg_kin, d_rsi_off, d_res, out_words, cap_words,
                               make_geom(c, true).segs_per_wave, d_carry);
        else {
            if (d_rsi_off) (void)hipMemsetAsync(d_rsi_off, 0, sizeof(uint64_t), st);   // empty batch: single entry
            if (out_cap >= 16) (void)hipMemsetAsync(d_out, 0, 16, st);
        }
        if (d_seg_table && nseg)
            hipLaunchKernelGGL(k_seg_table, dim3((uint32_t)((nseg + 255) / 256)), dim3(256), 0, st, c, d_in,
                               ws.seg_start, d_seg_table);
        mark(2);
        mark(3);   // (the buffer is no longer cleared: k_scan_apply zeroes the few shared words)
        if (nseg) dispatch(true, c, d_in, ws, out_words, cap_words, fast_ok, st);
        mark(4);
    }
}

// c describes the concatenation of n_chunks chunks of segs_per_chunk segments each (whole RSIs)
bool batch_uniform_ok(const Cfg &c, uint64_t segs_per_chunk)
{
    if (segs_per_chunk == 0 || segs_per_chunk > kScanChunk || c.total_segs % segs_per_chunk) return false;
    return segs_per_chunk % make_geom(c, true).segs_per_wave == 0;       // no wavefront works across a chunk border
}

void launch_encode_uniform_batch(const Cfg &c, const uint8_t *d_in, uint64_t segs_per_chunk, uint8_t *d_out,
                                 size_t out_cap, const EncWorkspace &ws, BatchChunk *d_chunks, EncResult *d_res,
                                 hipStream_t st)
{
    uint32_t *out_words = reinterpret_cast<uint32_t *>(d_out);
    const uint64_t cap_words = out_cap / 4;
    const uint32_t n = (uint32_t)(c.total_segs / segs_per_chunk);
    const uint32_t fast_ok = ((reinterpret_cast<uintptr_t>(d_in) & 15u) == 0 &&
                              ((uint64_t)c.rsi * c.bs * c.bytes) % 16 == 0) ? 1u : 0u;
    dispatch(false, c, d_in, ws, nullptr, 0, fast_ok, st);
    hipLaunchKernelGGL(k_batch_reduce, dim3(n), dim3(256), 0, st, ws.seg_bits, (uint32_t)segs_per_chunk, d_chunks);
    hipLaunchKernelGGL(k_batch_bases, dim3(1), dim3(256), 0, st, d_chunks, (uint64_t)n, (uint64_t)out_cap, d_res);
    hipLaunchKernelGGL(k_batch_apply, dim3(n), dim3(256), 0, st, ws.seg_bits, ws.seg_clamp, (uint32_t)segs_per_chunk,
                       d_chunks, ws.seg_start, ws.seg_kin, out_words, cap_words, make_geom(c, true).segs_per_wave);
    dispatch(true, c, d_in, ws, out_words, cap_words, fast_ok, st);
}

#endif      // AEC_ENC_PART

}  // namespace aec
