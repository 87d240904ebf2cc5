// aec_gpu.hip -- device-resident batch API (include/aec_gpu.h): context, workspace, enqueue.
// No reference counterpart; this is the host-side seam between the libaec ABI front-end
// (aec_abi.cpp) / GPU-resident callers and the kernels in aec_enc.hip / aec_dec.hip.
#include <hip/hip_runtime.h>

#include <new>

#include "../../include/aec_gpu.h"
#include "aec_cfg.h"
#include <cstdio>
#include <cstdlib>

#include "aec_kernels.h"

using namespace aec;

struct aec_gpu_ctx {
    int device;
    void *ws;          // encoder workspace
    size_t ws_bytes;
    bool profiling;    // record PhaseEvents around the kernels of the next calls
    // One event set per call, in a ring, so that a caller can time every launch of a loop without
    // synchronising inside it; aec_gpu_phase_ms() averages over the sets recorded since profiling
    // was switched on (the last kProfRing of them).
    static constexpr unsigned kProfRing = 32;
    PhaseEvents ev[kProfRing];
    unsigned enc_calls, dec_calls;
    // an emit-only call continues the set its plan call opened (its "scan" then spans the host's
    // plan exchange between the two calls)
    const PhaseEvents *enc_events(uint32_t phases)
    {
        if (!profiling) return nullptr;
        if ((phases & ENC_PLAN) || enc_calls == 0) enc_calls++;
        return &ev[(enc_calls - 1) % kProfRing];
    }
    const PhaseEvents *dec_events() { return profiling ? &ev[dec_calls++ % kProfRing] : nullptr; }
    SegEntry *seg_table;   // where the next encode / emit calls also leave the segment table (or null)
    void *idx_ws;          // speculative index tables (aec_idx.hip), grown on demand
    size_t idx_ws_bytes;
    uint64_t idx_hint;     // caller's estimate of the coded RSI size in bits (0 = derive from max_rsi)
    uint64_t idx_used_hint;  // ... and what the last index pass worked with (sizes the rings of the decode behind it)
    bool seg_filled;       // the last index pass with a segment-start table filled it (it ran over the trunk tables)
    uint64_t idx_stop_near;  // the next index pass is a piece of a longer stream (aec_gpu_set_index_piece)
    void *dec_ws;          // workspace of the segment-wise decode of bare streams (aec_dec.hip: launch_decode_bare)
    size_t dec_ws_bytes;
    ShardCarry *carry;     // device record: what precedes this context's shard (emit_planned)
    void *fused;           // control block of the single-pass encoder (ticket, fail flag, look-back granules)
    size_t fused_bytes;
};
static_assert(sizeof(aec_gpu_seg_entry) == sizeof(SegEntry), "segment table layout");

static_assert(sizeof(aec_gpu_enc_result) == sizeof(EncResult), "result layout");
static_assert(sizeof(aec_gpu_dec_result) == sizeof(DecResult), "result layout");

static int cfg_from(const aec_gpu_params *p, size_t in_bytes, bool enc, Cfg *c)
{
    return make_cfg(p->bits_per_sample, p->block_size, p->rsi, p->flags, in_bytes, enc, c);
}

extern "C" {

int aec_gpu_create(aec_gpu_ctx **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RC_MEM_ERROR;
    aec_gpu_ctx *ctx = new (std::nothrow) aec_gpu_ctx;
    if (!ctx) return RC_MEM_ERROR;
    ctx->device = dev;
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->profiling = false;
    ctx->seg_table = nullptr;
    ctx->idx_ws = nullptr;
    ctx->idx_ws_bytes = 0;
    ctx->idx_hint = 0;
    ctx->idx_used_hint = 0;
    ctx->seg_filled = false;
    ctx->idx_stop_near = 0;
    ctx->dec_ws = nullptr;
    ctx->dec_ws_bytes = 0;
    ctx->carry = nullptr;
    ctx->fused = nullptr;
    ctx->fused_bytes = 0;
    ctx->enc_calls = ctx->dec_calls = 0;
    for (auto &set : ctx->ev)
        for (auto &e : set.ev) e = nullptr;
    *out = ctx;
    return RC_OK;
}

void aec_gpu_destroy(aec_gpu_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->idx_ws) (void)hipFree(ctx->idx_ws);
    if (ctx->dec_ws) (void)hipFree(ctx->dec_ws);
    if (ctx->carry) (void)hipFree(ctx->carry);
    if (ctx->fused) (void)hipFree(ctx->fused);
    for (auto &set : ctx->ev)
        for (auto &e : set.ev)
            if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int aec_gpu_check_params(const aec_gpu_params *p, int for_encode)
{
    Cfg c;
    return cfg_from(p, 0, for_encode != 0, &c);
}

size_t aec_gpu_encode_bound(const aec_gpu_params *p, size_t in_bytes)
{
    Cfg c;
    if (cfg_from(p, in_bytes, true, &c) != RC_OK) return 0;
    return (max_encoded_bytes(c) + 15 + 16) & ~(size_t)15;
}

uint64_t aec_gpu_rsi_count(const aec_gpu_params *p, size_t in_bytes)
{
    Cfg c;
    if (cfg_from(p, in_bytes, true, &c) != RC_OK) return 0;
    return c.rsi_count;
}

uint64_t aec_gpu_block_count(const aec_gpu_params *p, size_t in_bytes)
{
    Cfg c;
    if (cfg_from(p, in_bytes, true, &c) != RC_OK) return 0;
    return c.total_blocks;
}

// workspace of the two-pass encoder (plan / emit: per-block summaries, segment scan)
static int reserve_two_pass(aec_gpu_ctx *ctx, const Cfg &c)
{
    size_t o[6];
    const size_t need = enc_workspace_bytes(c, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5]);
    if (need > ctx->ws_bytes) {
        if (ctx->ws) (void)hipFree(ctx->ws);
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        if (hipMalloc(&ctx->ws, need) != hipSuccess) return RC_MEM_ERROR;
        ctx->ws_bytes = need;
    }
    return RC_OK;
}

// control block of the single-pass encoder (16 bytes per 4..16 segments)
static int reserve_fused(aec_gpu_ctx *ctx, const Cfg &c)
{
    const size_t need = fused_ctl_bytes(c);
    if (need > ctx->fused_bytes) {
        if (ctx->fused) (void)hipFree(ctx->fused);
        ctx->fused = nullptr;
        ctx->fused_bytes = 0;
        const size_t want = need + need / 16;
        if (hipMalloc(&ctx->fused, want) != hipSuccess) return RC_MEM_ERROR;
        ctx->fused_bytes = want;
    }
    return RC_OK;
}

int aec_gpu_reserve(aec_gpu_ctx *ctx, const aec_gpu_params *p, size_t in_bytes)
{
    Cfg c;
    const int rc = cfg_from(p, in_bytes, true, &c);
    if (rc != RC_OK) return rc;
    // what aec_gpu_encode_async needs; the (much larger) workspace of the plan / emit pair is
    // allocated by the first plan call
    return fused_supported(c) ? reserve_fused(ctx, c) : reserve_two_pass(ctx, c);
}

static int encode_phases(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                         void *d_out, size_t out_cap, unsigned int start_bit, unsigned int k_in,
                         uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result, void *stream, uint32_t phases,
                         const ShardCarry *d_carry = nullptr)
{
    Cfg c;
    int rc = cfg_from(p, in_bytes, true, &c);
    if (rc != RC_OK) return rc;
    if (start_bit > 7 || k_in > 31) return RC_CONF_ERROR;
    if ((phases & ENC_EMIT) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) || (out_cap & 15u) || out_cap < 16))
        return RC_CONF_ERROR;
    const bool fused = phases == ENC_ALL && !d_carry && fused_supported(c);
    rc = fused ? reserve_fused(ctx, c) : reserve_two_pass(ctx, c);
    if (rc != RC_OK) return rc;
    (void)hipGetLastError();   // only launch errors of THIS call are reported below
    EncWorkspace ws{};
    if (fused) {
        ws.fused_ctl = ctx->fused;
    } else {
        size_t o[6];
        enc_workspace_bytes(c, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5]);
        uint8_t *base = static_cast<uint8_t *>(ctx->ws);
        ws.meta = reinterpret_cast<uint32_t *>(base + o[0]);
        ws.seg_bits = reinterpret_cast<uint32_t *>(base + o[1]);
        ws.seg_clamp = reinterpret_cast<uint16_t *>(base + o[2]);
        ws.seg_start = reinterpret_cast<uint64_t *>(base + o[3]);
        ws.seg_kin = base + o[4];
        ws.partials = reinterpret_cast<ScanPartial *>(base + o[5]);
        ws.fused_ctl = nullptr;
    }
    launch_encode(c, static_cast<const uint8_t *>(d_in), static_cast<uint8_t *>(d_out), out_cap, start_bit,
                  k_in, ws, d_rsi_bit_offsets, reinterpret_cast<EncResult *>(d_result),
                  static_cast<hipStream_t>(stream), ctx->enc_events(phases), phases, ctx->seg_table, d_carry);
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_encode_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                         void *d_out, size_t out_cap, unsigned int start_bit, unsigned int k_in,
                         uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result, void *stream)
{
    return encode_phases(ctx, p, d_in, in_bytes, d_out, out_cap, start_bit, k_in, d_rsi_bit_offsets, d_result,
                         stream, ENC_ALL);
}

int aec_gpu_encode_plan_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                              aec_gpu_enc_result *d_result, void *stream)
{
    return encode_phases(ctx, p, d_in, in_bytes, nullptr, 0, 0, 0, nullptr, d_result, stream, ENC_PLAN);
}

int aec_gpu_encode_emit_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                              void *d_out, size_t out_cap, unsigned int start_bit, unsigned int k_in,
                              uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result, void *stream)
{
    return encode_phases(ctx, p, d_in, in_bytes, d_out, out_cap, start_bit, k_in, d_rsi_bit_offsets, d_result,
                         stream, ENC_EMIT);
}

int aec_gpu_encode_emit_planned_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                                      void *d_out, size_t out_cap, const aec_gpu_enc_result *d_plans,
                                      unsigned int rank, uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result,
                                      void *stream)
{
    if (!d_plans || rank >= 64) return RC_CONF_ERROR;
    if (!ctx->carry && hipMalloc(reinterpret_cast<void **>(&ctx->carry), sizeof(ShardCarry)) != hipSuccess) {
        ctx->carry = nullptr;
        (void)hipGetLastError();
        return RC_MEM_ERROR;
    }
    launch_shard_carry(reinterpret_cast<const EncResult *>(d_plans), rank, ctx->carry,
                       static_cast<hipStream_t>(stream));
    return encode_phases(ctx, p, d_in, in_bytes, d_out, out_cap, 0, 0, d_rsi_bit_offsets, d_result, stream,
                         ENC_EMIT, ctx->carry);
}

int aec_gpu_stitch_async(const void *d_gathered, size_t slot_bytes, const aec_gpu_enc_result *d_plans,
                         unsigned int world, void *d_stream, size_t stream_cap, uint64_t *d_total_bytes,
                         void *stream)
{
    if (!d_gathered || !d_plans || !d_stream || world == 0 || world > 64 || (slot_bytes & 15u) ||
        (reinterpret_cast<uintptr_t>(d_gathered) & 15u) || (reinterpret_cast<uintptr_t>(d_stream) & 15u))
        return RC_CONF_ERROR;
    (void)hipGetLastError();
    launch_stitch(static_cast<const uint8_t *>(d_gathered), slot_bytes, reinterpret_cast<const EncResult *>(d_plans),
                  world, static_cast<uint8_t *>(d_stream), stream_cap, d_total_bytes,
                  static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_decode_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                         const uint64_t *d_rsi_bit_offsets, uint64_t n_rsi, uint64_t total_blocks,
                         void *d_out, aec_gpu_dec_result *d_result, void *stream)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if (reinterpret_cast<uintptr_t>(d_in) & 3u) return RC_CONF_ERROR;
    if (n_rsi && (total_blocks > n_rsi * c.rsi || total_blocks <= (n_rsi - 1) * c.rsi)) return RC_CONF_ERROR;
    (void)hipGetLastError();
    if (!launch_decode(c, static_cast<const uint8_t *>(d_in), in_bytes, d_rsi_bit_offsets, n_rsi, total_blocks,
                       static_cast<uint8_t *>(d_out), reinterpret_cast<DecResult *>(d_result),
                       static_cast<hipStream_t>(stream), ctx->dec_events()))
        return RC_MEM_ERROR;
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

uint64_t aec_gpu_segment_count(const aec_gpu_params *p, size_t in_bytes)
{
    Cfg c;
    if (cfg_from(p, in_bytes, true, &c) != RC_OK) return 0;
    return c.total_segs;
}

void aec_gpu_set_segment_table(aec_gpu_ctx *ctx, aec_gpu_seg_entry *d_table)
{
    ctx->seg_table = reinterpret_cast<SegEntry *>(d_table);
}

int aec_gpu_decode_segments_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                                  const aec_gpu_seg_entry *d_seg_table, uint64_t n_seg, uint64_t total_blocks,
                                  void *d_out, aec_gpu_dec_result *d_result, void *stream)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if (reinterpret_cast<uintptr_t>(d_in) & 3u) return RC_CONF_ERROR;
    // n_seg must be the segment count of total_blocks
    const uint64_t full = total_blocks / c.rsi, rem = total_blocks % c.rsi;
    if (n_seg != full * c.segs_per_rsi + (rem + 63) / 64) return RC_CONF_ERROR;
    (void)hipGetLastError();
    if (!launch_decode_segments(c, static_cast<const uint8_t *>(d_in), in_bytes,
                                reinterpret_cast<const SegEntry *>(d_seg_table), n_seg, total_blocks,
                                static_cast<uint8_t *>(d_out), reinterpret_cast<DecResult *>(d_result),
                                static_cast<hipStream_t>(stream), ctx->dec_events()))
        return RC_MEM_ERROR;
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

static int index_common(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                        uint64_t start_bit, uint32_t start_block, uint64_t rsi_start_bit,
                        uint64_t *d_rsi_bit_offsets, uint64_t max_rsi, aec_gpu_dec_result *d_result, void *stream,
                        uint32_t tail_slot, uint64_t *d_seg_bits = nullptr)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if (reinterpret_cast<uintptr_t>(d_in) & 3u) return RC_CONF_ERROR;
    if (start_block >= c.rsi || (start_block && rsi_start_bit > start_bit)) return RC_CONF_ERROR;
    (void)hipGetLastError();
    // Table workspace of the trunk index, sized from the average coded RSI: the caller's hint, else input
    // bits / expected RSIs.  A failed allocation only means the serial walk.
    const uint64_t in_bits = (uint64_t)in_bytes * 8;
    const uint64_t hint = ctx->idx_hint ? ctx->idx_hint
                          : ((max_rsi && in_bits > start_bit) ? (in_bits - start_bit) / max_rsi : 0);
    const size_t need = index_workspace_bytes(c, in_bytes, start_bit, hint);
    if (need > ctx->idx_ws_bytes) {
        if (ctx->idx_ws) (void)hipFree(ctx->idx_ws);     // (synchronises: no walker still reads it)
        ctx->idx_ws = nullptr;
        ctx->idx_ws_bytes = 0;
        const size_t want = need + need / 16;
        if (hipMalloc(&ctx->idx_ws, want) == hipSuccess) {
            ctx->idx_ws_bytes = want;
        } else {
            // (the every-bit scheme's tables are the largest by far for what they serve -- up to 270 MB for a 2 MiB
            // stream: without them before the serial walk)
            (void)hipGetLastError();
            const size_t less = index_workspace_bytes_large(c, in_bytes, start_bit, hint);
            if (less && less < need && hipMalloc(&ctx->idx_ws, less) == hipSuccess) ctx->idx_ws_bytes = less;
            else (void)hipGetLastError();
        }
    }
    ctx->idx_used_hint = hint;
    if (d_seg_bits && !decode_bare_supported(c)) d_seg_bits = nullptr;
    if (d_seg_bits && hipMemsetAsync(d_seg_bits, 0xFF, (size_t)(max_rsi + 1) * c.segs_per_rsi * 8, static_cast<hipStream_t>(stream)) != hipSuccess)
        return RC_MEM_ERROR;
    ctx->seg_filled = launch_index(c, static_cast<const uint8_t *>(d_in), in_bytes, start_bit, d_rsi_bit_offsets, max_rsi,
                                   reinterpret_cast<DecResult *>(d_result), static_cast<hipStream_t>(stream), ctx->idx_ws,
                                   ctx->idx_ws_bytes, hint, start_block, rsi_start_bit, tail_slot, d_seg_bits, ctx->idx_stop_near);
    ctx->idx_stop_near = 0;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && getenv("AEC_ABI_TRACE"))
        fprintf(stderr, "aec_gpu_index_async: %s (in_bytes %zu start %llu max_rsi %llu hint %llu ws %zu)\n",
                hipGetErrorString(e), in_bytes, (unsigned long long)start_bit, (unsigned long long)max_rsi,
                (unsigned long long)hint, ctx->idx_ws_bytes);
    return e == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_index_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                        uint64_t start_bit, uint64_t *d_rsi_bit_offsets, uint64_t max_rsi,
                        aec_gpu_dec_result *d_result, void *stream)
{
    return index_common(ctx, p, d_in, in_bytes, start_bit, 0u, start_bit, d_rsi_bit_offsets, max_rsi, d_result,
                        stream, 0u);
}

int aec_gpu_index_resume_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                               uint64_t start_bit, unsigned int start_block, uint64_t rsi_start_bit,
                               uint64_t *d_rsi_bit_offsets, uint64_t max_rsi, aec_gpu_dec_result *d_result,
                               void *stream)
{
    return index_common(ctx, p, d_in, in_bytes, start_bit, start_block, rsi_start_bit, d_rsi_bit_offsets, max_rsi,
                        d_result, stream, 1u);
}

int aec_gpu_decode_indexed_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                                 const uint64_t *d_rsi_bit_offsets, uint64_t max_rsi,
                                 const aec_gpu_dec_result *d_index_result, void *d_out,
                                 aec_gpu_dec_result *d_result, void *stream)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) || !d_index_result || d_index_result == d_result)
        return RC_CONF_ERROR;
    (void)hipGetLastError();
    if (!launch_decode(c, static_cast<const uint8_t *>(d_in), in_bytes, d_rsi_bit_offsets, max_rsi,
                       max_rsi * c.rsi, static_cast<uint8_t *>(d_out), reinterpret_cast<DecResult *>(d_result),
                       static_cast<hipStream_t>(stream), ctx->dec_events(),
                       reinterpret_cast<const DecResult *>(d_index_result)))
        return RC_MEM_ERROR;
    launch_decode_partial(c, static_cast<const uint8_t *>(d_in), in_bytes,
                          reinterpret_cast<const DecResult *>(d_index_result), static_cast<uint8_t *>(d_out),
                          reinterpret_cast<DecResult *>(d_result), static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

unsigned int aec_gpu_segments_per_rsi(const aec_gpu_params *p)
{
    Cfg c;
    return cfg_from(p, 0, false, &c) == RC_OK ? c.segs_per_rsi : 0u;
}

int aec_gpu_index_segments_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                                 uint64_t start_bit, unsigned int start_block, uint64_t rsi_start_bit,
                                 uint64_t *d_rsi_bit_offsets, uint64_t *d_seg_bits, uint64_t max_rsi,
                                 aec_gpu_dec_result *d_result, void *stream)
{
    return index_common(ctx, p, d_in, in_bytes, start_bit, start_block, start_block ? rsi_start_bit : start_bit,
                        d_rsi_bit_offsets, max_rsi, d_result, stream, 1u, d_seg_bits);
}

int aec_gpu_decode_bare_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                              const uint64_t *d_rsi_bit_offsets, const uint64_t *d_seg_bits, uint64_t max_rsi,
                              uint64_t total_blocks, const aec_gpu_dec_result *d_index_result, void *d_out,
                              aec_gpu_dec_result *d_result, void *stream)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) || d_index_result == d_result) return RC_CONF_ERROR;
    (void)hipGetLastError();
    // (segment starts the last index pass of this context did not fill are no use: every RSI by one lane then)
    const bool seg = d_seg_bits && ctx->seg_filled && decode_bare_supported(c);
    if (seg) {
        const size_t need = decode_bare_workspace_bytes(c, max_rsi);
        if (need > ctx->dec_ws_bytes) {
            if (ctx->dec_ws) (void)hipFree(ctx->dec_ws);
            ctx->dec_ws = nullptr;
            ctx->dec_ws_bytes = 0;
            if (hipMalloc(&ctx->dec_ws, need + need / 8) == hipSuccess) ctx->dec_ws_bytes = need + need / 8;
            else (void)hipGetLastError();
        }
    }
    const DecResult *idx = reinterpret_cast<const DecResult *>(d_index_result);
    if (!launch_decode_bare(c, static_cast<const uint8_t *>(d_in), in_bytes, d_rsi_bit_offsets, seg ? d_seg_bits : nullptr,
                            max_rsi, idx ? max_rsi * c.rsi : total_blocks, static_cast<uint8_t *>(d_out),
                            reinterpret_cast<DecResult *>(d_result), static_cast<hipStream_t>(stream), ctx->dec_events(),
                            idx, ctx->dec_ws, ctx->dec_ws_bytes, ctx->idx_used_hint / c.rsi))
        return RC_MEM_ERROR;
    if (idx)
        launch_decode_partial(c, static_cast<const uint8_t *>(d_in), in_bytes, idx, static_cast<uint8_t *>(d_out),
                              reinterpret_cast<DecResult *>(d_result), static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

void aec_gpu_set_index_hint(aec_gpu_ctx *ctx, uint64_t rsi_bits) { ctx->idx_hint = rsi_bits; }
void aec_gpu_set_index_piece(aec_gpu_ctx *ctx, uint64_t stop_near_bits) { ctx->idx_stop_near = stop_near_bits; }

int aec_gpu_index_is_windowed(const aec_gpu_params *p, size_t in_bytes, uint64_t rsi_bits)
{
    Cfg c;
    return cfg_from(p, 0, false, &c) == RC_OK && index_is_windowed(c, in_bytes, rsi_bits) ? 1 : 0;
}

int aec_gpu_index_scheme(const aec_gpu_params *p, size_t in_bytes, uint64_t rsi_bits, unsigned int start_block)
{
    Cfg c;
    if (cfg_from(p, 0, false, &c) != RC_OK) return 0;
    return index_scheme(c, in_bytes, rsi_bits, start_block);
}

void aec_gpu_trim(aec_gpu_ctx *ctx, size_t keep_bytes)
{
    if (ctx->ws && ctx->ws_bytes > keep_bytes) {
        (void)hipFree(ctx->ws);
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    if (ctx->fused && ctx->fused_bytes > keep_bytes) {
        (void)hipFree(ctx->fused);
        ctx->fused = nullptr;
        ctx->fused_bytes = 0;
    }
    if (ctx->idx_ws && ctx->idx_ws_bytes > keep_bytes) {
        (void)hipFree(ctx->idx_ws);
        ctx->idx_ws = nullptr;
        ctx->idx_ws_bytes = 0;
    }
    if (ctx->dec_ws && ctx->dec_ws_bytes > keep_bytes) {
        (void)hipFree(ctx->dec_ws);
        ctx->dec_ws = nullptr;
        ctx->dec_ws_bytes = 0;
    }
}

size_t aec_gpu_held_bytes(const aec_gpu_ctx *ctx)
{
    return ctx ? ctx->ws_bytes + ctx->fused_bytes + ctx->idx_ws_bytes + ctx->dec_ws_bytes : 0;
}

int aec_gpu_index_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                              const uint64_t *d_chunk_offsets, uint64_t n_chunks, uint64_t rsi_per_chunk,
                              uint64_t *d_rsi_bit_offsets, aec_gpu_dec_result *d_results, void *stream)
{
    (void)ctx;
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(d_in) & 15u) || rsi_per_chunk == 0) return RC_CONF_ERROR;
    (void)hipGetLastError();
    launch_index_batch(c, static_cast<const uint8_t *>(d_in), in_bytes, d_chunk_offsets, n_chunks, rsi_per_chunk,
                       d_rsi_bit_offsets, reinterpret_cast<DecResult *>(d_results),
                       static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_batch_uses_tables(aec_gpu_ctx *ctx, const aec_gpu_params *p, size_t in_bytes, uint64_t n_chunks,
                              uint64_t rsi_per_chunk)
{
    Cfg c;
    if (cfg_from(p, 0, false, &c) != RC_OK || n_chunks == 0 || rsi_per_chunk == 0) return 0;
    const size_t mean = in_bytes / n_chunks;
    const uint64_t hint = ctx->idx_hint ? ctx->idx_hint : (uint64_t)mean * 8 / rsi_per_chunk;
    return mean >= ((size_t)16 << 10) && index_batch_workspace_bytes(c, in_bytes, n_chunks, mean * 2 + 65536, hint) != 0;
}

int aec_gpu_decode_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t in_bytes,
                               const uint64_t *d_chunk_offsets, uint64_t n_chunks, uint64_t rsi_per_chunk,
                               uint64_t *d_rsi_bit_offsets, void *d_out, aec_gpu_dec_result *d_results,
                               aec_gpu_dec_result *d_result, void *stream)
{
    Cfg c;
    const int rc = cfg_from(p, 0, false, &c);
    if (rc != RC_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(d_in) & 15u) || rsi_per_chunk == 0 || rsi_per_chunk > 0xFFFFFFFFull)
        return RC_CONF_ERROR;
    if (n_chunks == 0) return RC_OK;
    (void)hipGetLastError();
    // Streams long enough for the window tables (low-entropy data, tens of KiB and more per stream): ONE
    // speculation launch for the whole buffer, then every stream's wavefront hops over the tables.  The largest
    // stream is not known here: the mean, with room to spare, sizes the hop lists.
    const size_t mean = in_bytes / n_chunks, max_chunk = mean * 2 + 65536;
    const uint64_t hint = ctx->idx_hint ? ctx->idx_hint : (uint64_t)mean * 8 / rsi_per_chunk;
    size_t need = mean >= ((size_t)16 << 10) ? index_batch_workspace_bytes(c, in_bytes, n_chunks, max_chunk, hint) : 0;
    if (need > ctx->idx_ws_bytes) {
        if (ctx->idx_ws) (void)hipFree(ctx->idx_ws);
        ctx->idx_ws = nullptr;
        ctx->idx_ws_bytes = 0;
        if (hipMalloc(&ctx->idx_ws, need + need / 16) == hipSuccess) ctx->idx_ws_bytes = need + need / 16;
        else (void)hipGetLastError();
    }
    launch_index_batch(c, static_cast<const uint8_t *>(d_in), in_bytes, d_chunk_offsets, n_chunks, rsi_per_chunk,
                       d_rsi_bit_offsets, reinterpret_cast<DecResult *>(d_results), static_cast<hipStream_t>(stream),
                       need ? ctx->idx_ws : nullptr, ctx->idx_ws_bytes, max_chunk, hint);
    if (!launch_decode(c, static_cast<const uint8_t *>(d_in), in_bytes, d_rsi_bit_offsets, n_chunks * rsi_per_chunk,
                       n_chunks * rsi_per_chunk * c.rsi, static_cast<uint8_t *>(d_out),
                       reinterpret_cast<DecResult *>(d_result), static_cast<hipStream_t>(stream), ctx->dec_events(),
                       nullptr, reinterpret_cast<const DecResult *>(d_results), (uint32_t)rsi_per_chunk))
        return RC_MEM_ERROR;
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_encode_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                               const uint64_t *chunk_offsets_host, uint64_t n_chunks, void *d_out, size_t slot_bytes,
                               aec_gpu_enc_result *d_results, void *stream)
{
    if (!chunk_offsets_host || (slot_bytes & 15u) || (reinterpret_cast<uintptr_t>(d_out) & 15u)) return RC_CONF_ERROR;
    // every chunk is a stream of its own (k = 0, bit 0 of its slot): the chunks go through the encoder
    // kernels one after the other on `stream`, sharing the context's workspace; nothing comes back to the host
    for (uint64_t i = 0; i < n_chunks; i++) {
        const uint64_t lo = chunk_offsets_host[i], hi = chunk_offsets_host[i + 1];
        if (hi < lo || (lo & 15u)) return RC_CONF_ERROR;
        const int rc = encode_phases(ctx, p, static_cast<const uint8_t *>(d_in) + lo, (size_t)(hi - lo),
                                     static_cast<uint8_t *>(d_out) + i * slot_bytes, slot_bytes, 0, 0, nullptr,
                                     d_results + i, stream, ENC_ALL);
        if (rc != RC_OK) return rc;
    }
    return RC_OK;
}

int aec_gpu_encode_uniform_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in, size_t chunk_bytes,
                                       uint64_t n_chunks, void *d_out, size_t out_cap, aec_gpu_batch_chunk *d_chunks,
                                       aec_gpu_enc_result *d_result, void *stream)
{
    static_assert(sizeof(aec_gpu_batch_chunk) == sizeof(BatchChunk), "public mirror of BatchChunk");
    if (!n_chunks || !chunk_bytes || (reinterpret_cast<uintptr_t>(d_out) & 15u) || (out_cap & 15u) || out_cap < 16)
        return RC_CONF_ERROR;
    Cfg one, c;
    int rc = cfg_from(p, chunk_bytes, true, &one);
    if (rc != RC_OK) return rc;
    const size_t rsi_bytes = (size_t)one.rsi * one.bs * one.bytes;
    if (chunk_bytes % rsi_bytes) return RC_CONF_ERROR;                       // whole RSIs only
    rc = cfg_from(p, chunk_bytes * n_chunks, true, &c);
    if (rc != RC_OK) return rc;
    if (!batch_uniform_ok(c, one.total_segs)) return RC_CONF_ERROR;          // (aec_gpu_uniform_batch_ok says so beforehand)
    rc = reserve_two_pass(ctx, c);
    if (rc != RC_OK) return rc;
    (void)hipGetLastError();
    size_t o[6];
    enc_workspace_bytes(c, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5]);
    uint8_t *base = static_cast<uint8_t *>(ctx->ws);
    EncWorkspace ws{};
    ws.meta = reinterpret_cast<uint32_t *>(base + o[0]);
    ws.seg_bits = reinterpret_cast<uint32_t *>(base + o[1]);
    ws.seg_clamp = reinterpret_cast<uint16_t *>(base + o[2]);
    ws.seg_start = reinterpret_cast<uint64_t *>(base + o[3]);
    ws.seg_kin = base + o[4];
    ws.partials = reinterpret_cast<ScanPartial *>(base + o[5]);
    launch_encode_uniform_batch(c, static_cast<const uint8_t *>(d_in), one.total_segs, static_cast<uint8_t *>(d_out),
                                out_cap, ws, reinterpret_cast<BatchChunk *>(d_chunks),
                                reinterpret_cast<EncResult *>(d_result), static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? RC_OK : RC_MEM_ERROR;
}

int aec_gpu_uniform_batch_ok(const aec_gpu_params *p, size_t chunk_bytes, uint64_t n_chunks)
{
    Cfg one, c;
    if (!n_chunks || !chunk_bytes || cfg_from(p, chunk_bytes, true, &one) != RC_OK) return 0;
    if (chunk_bytes % ((size_t)one.rsi * one.bs * one.bytes)) return 0;
    if (cfg_from(p, chunk_bytes * n_chunks, true, &c) != RC_OK) return 0;
    return batch_uniform_ok(c, one.total_segs) ? 1 : 0;
}

int aec_gpu_profile(aec_gpu_ctx *ctx, int enable)
{
    if (enable) {
        for (auto &set : ctx->ev)
            for (auto &e : set.ev)
                if (!e && hipEventCreate(&e) != hipSuccess) return RC_MEM_ERROR;
        ctx->enc_calls = ctx->dec_calls = 0;
    }
    ctx->profiling = enable != 0;
    return RC_OK;
}

int aec_gpu_phase_ms(aec_gpu_ctx *ctx, float *ms /*[5]: analyze, scan, clear, pack, decode*/)
{
    static const int pair[5][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {5, 6}};
    for (int i = 0; i < 5; i++) {
        const unsigned calls = i < 4 ? ctx->enc_calls : ctx->dec_calls;
        const unsigned sets = calls < aec_gpu_ctx::kProfRing ? calls : aec_gpu_ctx::kProfRing;
        double sum = 0;
        unsigned got = 0;
        for (unsigned s = 0; s < sets; s++) {
            hipEvent_t a = ctx->ev[s].ev[pair[i][0]], b = ctx->ev[s].ev[pair[i][1]];
            if (!a || !b || hipEventSynchronize(b) != hipSuccess) continue;
            float t = 0;
            if (hipEventElapsedTime(&t, a, b) == hipSuccess) { sum += t; got++; }
        }
        ms[i] = got ? static_cast<float>(sum / got) : -1.0f;
    }
    return RC_OK;
}

}  // extern "C"
