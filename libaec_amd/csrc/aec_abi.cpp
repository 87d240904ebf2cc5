// aec_abi.cpp -- the libaec C ABI (include/libaec.h) on top of the device-resident batch API.
//
// This is the host-side stream layer the reference implements as two resumable finite state
// machines (reference src/encode.c:467-518, 661-754, 909-963 and src/decode.c:342-400,
// 797-854).  Its contract is kept -- any chunking of input and output, cumulative
// total_in/total_out, AEC_FLUSH semantics, the same return codes -- but the states are replaced
// by staging and batching:
//
//   encode  input is collected on the host until it is worth a launch (a batch threshold, AEC_FLUSH,
//           or a call that brings no new input while whole RSIs wait); every whole RSI staged is then
//           coded in ONE batch on the GPU with the bit position and k carried between batches.  Cost
//           is linear in the input for any chunking (a caller feeding one sample per call pays a
//           vector append per call, not a launch).
//   decode  the compressed stream is kept RESIDENT ON THE DEVICE: only new bytes are uploaded, the
//           index walker resumes at the coded data set where it stopped (position + blocks of the
//           current RSI), the decoder takes its item counts from the walker's record on the device,
//           and one host synchronisation returns both records and the first output bytes.  A batch
//           is bounded by the room the caller offers (at least kMinBatchOut), so memory is
//           O(offered output + undecoded input), never O(decoded size of everything staged).
//
// There is no CPU codec in here: without a working HIP device every call fails with AEC_MEM_ERROR.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/aec_gpu.h"
#include "../../include/libaec.h"
#include "aec_cfg.h"
#include "aec_pool.h"

using namespace aec;

namespace {

// AEC_ABI_TRACE=1 in the environment (the one variable the product library reads): say on stderr where a
// device-side failure was detected, and one line per decode batch (span, hint, what the walker brought back)
bool trace_on()
{
    static const bool trace = getenv("AEC_ABI_TRACE") != nullptr;
    return trace;
}
int fail_at(int code, int line)
{
    if (trace_on()) fprintf(stderr, "libaec (MI355X): error %d raised at aec_abi.cpp:%d (last HIP error: %s)\n", code, line,
                       hipGetErrorString(hipPeekAtLastError()));
    return code;
}
#define AEC_FAIL(code) fail_at((code), __LINE__)

// What the index pass is told about the coded RSIs: a look-ahead of 1.5 means (the window tables are sized from it) --
// except for RSIs of uncompressed blocks, where the pass is told the mean itself so that it leaves out the schemes that
// look for reference samples (aec_kernels.h: index_incompressible), and a look-ahead that happens to fall into that
// band is moved beyond it.
uint64_t index_hint_of(const Cfg &c, uint64_t mean)
{
    if (index_incompressible(c, mean)) return mean;
    const uint64_t h = mean + mean / 2;
    return index_incompressible(c, h) ? (uint64_t)c.rsi * (c.id_len + (uint64_t)c.bs * c.bps) + 65 : h;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // keep: bytes at the front that must survive a reallocation (copied device to device)
    bool ensure(size_t n, size_t keep = 0)
    {
        if (n <= cap) return true;
        size_t want = n + n / 4 + 256;
        want = (want + 255) & ~(size_t)255;
        void *q = nullptr;
        if (hipMalloc(&q, want) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (p && keep && hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipFree(q);
            return false;
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return true;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// batching thresholds
constexpr size_t kEncBatchBytes = (size_t)1 << 20;   // staged input that is worth a launch without AEC_FLUSH
constexpr size_t kEncDirectMin = (size_t)64 << 10;   // whole RSIs offered in one call: coded from the caller's buffer
constexpr size_t kDecTinyCall = 8;                   // a call that brings at most this many bytes is a trickle ...
constexpr size_t kDecTrickle = 4096;                 // ... collected on the host up to this many before a launch
constexpr size_t kDecDirectMin = 4096;               // input of at least this size goes straight to the device
constexpr size_t kMinBatchOut = (size_t)4 << 20;     // output a decode batch may produce beyond the room offered
constexpr size_t kBacklogMax = (size_t)64 << 20;     // undecoded input held on the device before more is accepted
constexpr size_t kBounce = (size_t)256 << 10;        // pinned bounce buffer: first output bytes ride with the records
constexpr size_t kPipeOut = (size_t)64 << 20;        // decode: output per batch where batches are pipelined (below)
constexpr size_t kAsyncMin = (size_t)1 << 20;        // ... and the smallest copy-out that is worth the side stream

}  // namespace

struct internal_state {
    bool encoder;
    aec_gpu_params prm;
    Cfg cfg;                       // derived values (sizes are per batch, not used from here)
    int device;
    aec_gpu_ctx *ctx;
    hipStream_t stream;
    DevBuf d_in, d_out, d_off, d_res, d_seg;   // (d_seg: segment starts beside the RSI starts, long RSIs only)
    DevBuf d_out2;                 // decoder: second output buffer (batches alternate while copies are in flight)
    hipStream_t copy_stream;       // decoder: the copy-out of a batch runs here, beside the kernels of the next one
    hipEvent_t ev_copied[2];       // ... and this says when the copy out of buffer i has finished
    bool copy_pending;             // a copy to the caller's buffer is in flight: waited for before the call returns
    unsigned out_sel;              // which output buffer the next batch writes
    uint8_t *h_res;                // pinned: 256 bytes of records, then kBounce bytes of bounce buffer

    std::vector<uint8_t> stage;    // encoder: input not yet coded; decoder: input not yet on the device
    size_t stage_pos;              // encoder: first staged byte still to be coded
    std::vector<uint8_t> outq;     // produced bytes not yet delivered
    size_t outq_pos;

    // encoder carry (reference state->k, state->bits / *state->cds)
    uint32_t k;
    uint32_t part_bits;            // bits used in the open byte, 0..7
    uint8_t part_byte;
    bool any_bits;                 // at least one stream bit produced
    bool finished;                 // final byte queued
    int flush;                     // last flush argument
    bool flushed;                  // reference state->flushed

    // decoder: d_in holds stream bytes [base, base + d_len); all *_bit values are absolute stream bits
    uint64_t base;                 // multiple of 16
    size_t d_len;
    uint64_t rsi_start_bit;        // start of the RSI being decoded
    uint64_t rsi_bits_seen;        // average coded RSI of the previous batch (0 = none yet)
    uint64_t walk_bit;             // coded-data-set boundary where the index walker resumes
    uint32_t walk_blocks;          // blocks of that RSI in front of walk_bit
    uint64_t delivered;            // samples of that RSI already handed out
    size_t walked_len;             // d_len at the last index pass (anything beyond it is new)
    uint64_t span_mul;             // widening of the batch's input span (coded data sets beyond the encoder's bound)
    bool span_wide;                // pipelined batches: the tight span did not hold once, the worst case from now on
    bool more;                     // the last batch stopped at its RSI bound: decodable input remains
    bool launched;                 // at least one batch has run
    int sticky_error;
};

namespace {

// Device-side belongings of a stream.  Callers like the HDF5 SZIP filter run one
// aec_buffer_encode / aec_buffer_decode per chunk, i.e. an init / end cycle per megabyte: creating
// a HIP stream, a pinned result record and four device buffers each time (and freeing them, which
// synchronises the device) costs more than coding the chunk.  So *_end parks them in a small
// process-wide pool and *_init takes them from there.  Buffers above kKeepBytes are released
// first; the pool is never torn down at exit (the driver reclaims it with the process -- calling
// into HIP from static destructors is not safe).
struct Kit {
    int device = -1;
    aec_gpu_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    DevBuf d_in, d_out, d_off, d_res, d_seg, d_out2;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr};
    uint8_t *h_res = nullptr;
    uint8_t *h_stage = nullptr;    // pinned staging of the batch entry points (many chunks, one transfer)
    size_t h_stage_cap = 0;
};
// at most kPoolMax parked kits, none holding a buffer above kKeepBytes, all of them together at most kPoolBytes
constexpr size_t kPoolMax = 8, kKeepBytes = (size_t)256 << 20, kPoolBytes = (size_t)1 << 30;
constexpr size_t kStageKeep = (size_t)96 << 20, kStagePiece = (size_t)64 << 20, kPoolPinned = (size_t)192 << 20;
std::mutex g_pool_mu;
std::vector<Kit> *g_pool = nullptr;      // heap object on purpose: no destructor at exit

void destroy_kit(Kit &k)
{
    k.d_in.release();
    k.d_out.release();
    k.d_off.release();
    k.d_res.release();
    k.d_seg.release();
    k.d_out2.release();
    if (k.copy_stream) (void)hipStreamDestroy(k.copy_stream);
    for (hipEvent_t &e : k.ev_copied)
        if (e) (void)hipEventDestroy(e);
    if (k.h_res) (void)hipHostFree(k.h_res);
    if (k.h_stage) (void)hipHostFree(k.h_stage);
    if (k.stream) (void)hipStreamDestroy(k.stream);
    if (k.ctx) aec_gpu_destroy(k.ctx);
    k = Kit{};
}

bool take_kit(int device, Kit *out)
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (!g_pool) return false;
    for (size_t i = 0; i < g_pool->size(); i++)
        if ((*g_pool)[i].device == device) {
            *out = (*g_pool)[i];
            g_pool->erase(g_pool->begin() + (ptrdiff_t)i);
            return true;
        }
    return false;
}

size_t kit_bytes(const Kit &k)
{
    return k.d_in.cap + k.d_out.cap + k.d_off.cap + k.d_res.cap + k.d_seg.cap + k.d_out2.cap + aec_gpu_held_bytes(k.ctx);
}

void park_kit(Kit &k)
{
    // an error return may have left copies or kernels of this stream object enqueued: nothing of it may still
    // run when the next owner writes the buffers
    if (hipStreamSynchronize(k.stream) != hipSuccess || (k.copy_stream && hipStreamSynchronize(k.copy_stream) != hipSuccess)) {
        (void)hipGetLastError();
        destroy_kit(k);
        return;
    }
    for (DevBuf *b : {&k.d_in, &k.d_out, &k.d_off, &k.d_seg, &k.d_out2})
        if (b->cap > kKeepBytes) b->release();
    if (k.h_stage_cap > kStageKeep) {
        (void)hipHostFree(k.h_stage);
        k.h_stage = nullptr;
        k.h_stage_cap = 0;
    }
    aec_gpu_trim(k.ctx, kKeepBytes);
    aec_gpu_set_index_hint(k.ctx, 0);
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (!g_pool) g_pool = new (std::nothrow) std::vector<Kit>();
        size_t held = kit_bytes(k), pinned = k.h_stage_cap;
        if (g_pool)
            for (const Kit &o : *g_pool) {
                held += kit_bytes(o);
                pinned += o.h_stage_cap;
            }
        if (pinned > kPoolPinned && k.h_stage) {     // page-locked host memory of all parked kits together is bounded too
            (void)hipHostFree(k.h_stage);
            k.h_stage = nullptr;
            k.h_stage_cap = 0;
        }
        if (g_pool && g_pool->size() < kPoolMax && held <= kPoolBytes) {
            g_pool->push_back(k);
            k = Kit{};
            return;
        }
    }
    destroy_kit(k);
}

void free_state(internal_state *s)
{
    if (!s) return;
    Kit k;
    k.device = s->device;
    k.ctx = s->ctx;
    k.stream = s->stream;
    k.d_in = s->d_in;
    k.d_out = s->d_out;
    k.d_off = s->d_off;
    k.d_res = s->d_res;
    k.d_seg = s->d_seg;
    k.d_out2 = s->d_out2;
    k.copy_stream = s->copy_stream;
    k.ev_copied[0] = s->ev_copied[0];
    k.ev_copied[1] = s->ev_copied[1];
    k.h_res = s->h_res;
    if (k.ctx && k.stream && k.h_res && k.d_res.p) park_kit(k);
    else destroy_kit(k);
    delete s;
}

int init_common(struct aec_stream *strm, bool enc)
{
    aec_gpu_params prm{strm->bits_per_sample, strm->block_size, strm->rsi, strm->flags};
    Cfg c;
    const int rc = make_cfg(prm.bits_per_sample, prm.block_size, prm.rsi, prm.flags, 0, enc, &c);
    if (rc != RC_OK) return AEC_FAIL(rc);
    internal_state *s = new (std::nothrow) internal_state();
    if (!s) return AEC_FAIL(AEC_MEM_ERROR);
    s->encoder = enc;
    s->prm = prm;
    s->cfg = c;
    s->stage_pos = 0;
    s->outq_pos = 0;
    s->k = 0;
    s->part_bits = 0;
    s->part_byte = 0;
    s->any_bits = false;
    s->finished = false;
    s->flush = AEC_NO_FLUSH;
    s->flushed = false;
    s->base = 0;
    s->d_len = 0;
    s->rsi_start_bit = 0;
    s->rsi_bits_seen = 0;
    s->walk_bit = 0;
    s->walk_blocks = 0;
    s->delivered = 0;
    s->walked_len = 0;
    s->span_mul = 1;
    s->span_wide = false;
    s->more = false;
    s->launched = false;
    s->sticky_error = AEC_OK;
    s->ctx = nullptr;
    s->stream = nullptr;
    s->copy_stream = nullptr;
    s->ev_copied[0] = s->ev_copied[1] = nullptr;
    s->copy_pending = false;
    s->out_sel = 0;
    s->h_res = nullptr;
    s->device = -1;
    Kit k;
    if (hipGetDevice(&s->device) == hipSuccess && take_kit(s->device, &k)) {
        s->ctx = k.ctx;
        s->stream = k.stream;
        s->d_in = k.d_in;
        s->d_out = k.d_out;
        s->d_off = k.d_off;
        s->d_res = k.d_res;
        s->d_seg = k.d_seg;
        s->d_out2 = k.d_out2;
        s->copy_stream = k.copy_stream;
        s->ev_copied[0] = k.ev_copied[0];
        s->ev_copied[1] = k.ev_copied[1];
        s->h_res = k.h_res;
        aec_gpu_set_index_hint(s->ctx, 0);
    } else if (s->device < 0 || aec_gpu_create(&s->ctx) != RC_OK || hipStreamCreate(&s->stream) != hipSuccess ||
               hipHostMalloc(reinterpret_cast<void **>(&s->h_res), 256 + kBounce, hipHostMallocDefault) != hipSuccess ||
               !s->d_res.ensure(256)) {
        free_state(s);
        return AEC_FAIL(AEC_MEM_ERROR);   // no usable HIP device: the product has no CPU path
    }
    strm->state = s;
    strm->total_in = 0;         // reference encode.c:897-898, decode.c:785-786
    strm->total_out = 0;
    return AEC_OK;
}

inline size_t drain(struct aec_stream *strm, internal_state *s, size_t granule)
{
    size_t n = s->outq.size() - s->outq_pos;
    if (n == 0) return 0;
    if (n > strm->avail_out) n = strm->avail_out;
    n -= n % granule;
    if (n) {
        memcpy(strm->next_out, s->outq.data() + s->outq_pos, n);
        strm->next_out += n;
        strm->avail_out -= n;
        s->outq_pos += n;
    }
    if (s->outq_pos == s->outq.size()) {
        s->outq.clear();
        s->outq_pos = 0;
    }
    return n;
}

// Code `nbytes` of input (whole samples) as one GPU batch; append produced whole bytes to the queue
// and keep the open byte / k as carry.  Nothing is queued when this runs, so finished bytes go
// straight into the caller's buffer as far as it has room: the first kBounce of them ride with the
// result record (one synchronisation), only what does not fit -- and the open byte -- is queued.
int encode_batch(internal_state *s, const uint8_t *data, size_t nbytes, struct aec_stream *strm)
{
    if (nbytes == 0) return AEC_OK;
    const size_t cap = aec_gpu_encode_bound(&s->prm, nbytes);
    if (!s->d_in.ensure(nbytes + 16) || !s->d_out.ensure(cap)) return AEC_FAIL(AEC_MEM_ERROR);
    if (hipMemcpyAsync(s->d_in.p, data, nbytes, hipMemcpyHostToDevice, s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const int rc = aec_gpu_encode_async(s->ctx, &s->prm, s->d_in.p, nbytes, s->d_out.p, cap, s->part_bits,
                                        s->k, nullptr, static_cast<aec_gpu_enc_result *>(s->d_res.p),
                                        s->stream);
    if (rc != RC_OK) return AEC_FAIL(rc);
    const uint8_t *d_bytes = static_cast<const uint8_t *>(s->d_out.p);
    uint8_t *bounce = s->h_res + 256;
    const size_t spec = cap < kBounce ? cap : kBounce;          // speculative: the size is not known yet
    if (hipMemcpyAsync(s->h_res, s->d_res.p, sizeof(aec_gpu_enc_result), hipMemcpyDeviceToHost, s->stream) !=
            hipSuccess ||
        hipMemcpyAsync(bounce, d_bytes, spec, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const aec_gpu_enc_result res = *reinterpret_cast<aec_gpu_enc_result *>(s->h_res);
    if (res.overflow) return AEC_FAIL(AEC_MEM_ERROR);   // cannot happen: cap is the worst case
    const uint64_t bits = (uint64_t)s->part_bits + res.total_bits;
    const size_t whole = (size_t)(bits / 8), nb = (size_t)((bits + 7) / 8);
    // bytes [0, whole) are finished, byte `whole` (if nb > whole) is the new open byte
    size_t direct = whole < strm->avail_out ? whole : strm->avail_out;
    uint8_t *out0 = strm->next_out;
    if (direct) {
        const size_t from_bounce = direct < spec ? direct : spec;
        memcpy(strm->next_out, bounce, from_bounce);
        if (direct > from_bounce &&
            hipMemcpy(strm->next_out + from_bounce, d_bytes + from_bounce, direct - from_bounce,
                      hipMemcpyDeviceToHost) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
    }
    const size_t rest = nb - direct;                     // queued: unfinished bytes + the open byte
    const size_t at = s->outq.size();
    s->outq.resize(at + rest);
    if (rest) {
        size_t done = 0;
        if (direct < spec) {
            done = spec - direct < rest ? spec - direct : rest;
            memcpy(s->outq.data() + at, bounce + direct, done);
        }
        if (rest > done &&
            hipMemcpy(s->outq.data() + at + done, d_bytes + direct + done, rest - done, hipMemcpyDeviceToHost) !=
                hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
    }
    // the byte the batch started in carries the bits of the previous batch
    if (direct) out0[0] |= s->part_byte;
    else if (nb) s->outq[at] |= s->part_byte;
    s->part_bits = (uint32_t)(bits % 8);
    s->part_byte = s->part_bits ? s->outq[at + whole - direct] : 0;
    s->outq.resize(at + whole - direct);
    if (direct) {
        strm->next_out += direct;
        strm->avail_out -= direct;
    }
    s->k = res.k_out;
    if (res.total_bits) s->any_bits = true;
    return AEC_OK;
}

// ---- decoder ------------------------------------------------------------------------------------

// bytes one RSI can occupy at most in the stream (every block uncompressed)
inline uint64_t worst_rsi_bytes(const Cfg &c)
{
    return ((uint64_t)c.rsi * (c.id_len + (uint64_t)c.bs * c.bps) + c.bps + 7) / 8 + 1;
}

// Append `n` bytes at `src` (host) to the device-resident stream.
int upload(internal_state *s, const uint8_t *src, size_t n)
{
    if (n == 0) return AEC_OK;
    if (!s->d_in.ensure(s->d_len + n + 32, s->d_len)) return AEC_FAIL(AEC_MEM_ERROR);
    if (hipMemcpyAsync(static_cast<uint8_t *>(s->d_in.p) + s->d_len, src, n, hipMemcpyHostToDevice, s->stream) !=
        hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    s->d_len += n;
    return AEC_OK;
}

// One batch: index from where the walker stopped, decode what it found, hand out / queue the samples.
int decode_run(internal_state *s, struct aec_stream *strm)
{
    const Cfg &c = s->cfg;
    if (!s->stage.empty()) {
        const int rc = upload(s, s->stage.data(), s->stage.size());
        if (rc != AEC_OK) return rc;
        // (the copy out of this pageable vector has been staged by the runtime when the call returns,
        // and the batch below is waited for in any case)
        s->stage.clear();
    }
    s->launched = true;
    s->more = false;
    if (s->d_len == 0) return AEC_OK;
    const size_t blk_bytes = (size_t)c.bs * c.bytes;
    const size_t rsi_bytes = (size_t)c.rsi * blk_bytes;
    const uint64_t base_bits = s->base * 8;
    const uint64_t walk_rel = s->walk_bit - base_bits, rsi_rel = s->rsi_start_bit - base_bits;
    const size_t skip = (size_t)s->delivered * c.bytes;   // bytes of the current RSI already handed out
    const size_t want_out = strm->avail_out;              // (what the caller asks of THIS call)

    // Look-ahead of the speculative index = a small multiple of the average coded RSI: measured on
    // the previous batch of this stream, else estimated from the room the caller offers for output
    // (whole-buffer callers offer the decoded size).
    uint64_t hint = s->rsi_bits_seen;
    if (!hint && strm->avail_out >= rsi_bytes) {
        const uint64_t expect = (strm->avail_out + skip + rsi_bytes - 1) / rsi_bytes;
        hint = ((uint64_t)s->d_len * 8 - rsi_rel) / expect;
    }
    // Large outputs of streams whose index pass is cheap per call (the window tables): batches of kPipeOut, the copy
    // of one batch to the caller's buffer on a side stream beside the index pass and the decode of the next one (one
    // batch for everything was upload, index, decode, copy one behind the other: 21 GB/s for 256 MiB of config 2
    // against 42 for the encoder).  The trunk index pays too much per call for that (spans, burn-in): one batch.
    const bool pipe = strm->avail_out >= kPipeOut + kPipeOut / 2 &&
                      aec_gpu_index_is_windowed(&s->prm, s->d_len - (size_t)(rsi_rel / 8), hint + hint / 2) != 0;
    // bound of the batch: the room offered (at least kMinBatchOut), and no more than the input can hold
    const size_t room = (pipe ? kPipeOut : (strm->avail_out > kMinBatchOut ? strm->avail_out : kMinBatchOut)) + skip;
    uint64_t max_rsi = room / rsi_bytes + 2;
    const uint64_t min_rsi_bits = (uint64_t)c.segs_per_rsi * (c.id_len + 2) + ((c.flags & F_PREPROCESS) ? c.bps : 0);
    const uint64_t avail_bits = (uint64_t)s->d_len * 8 - rsi_rel;
    if (max_rsi > avail_bits / min_rsi_bits + 2) max_rsi = avail_bits / min_rsi_bits + 2;
    // the part of the resident stream this batch can need
    // (worst_rsi_bytes is what an ENCODER makes of an RSI at most; the format allows longer ones: should the
    // walker run out of input inside the span while more is resident, the next batch looks further)
    uint64_t span = walk_rel / 8 + max_rsi * worst_rsi_bytes(c) * s->span_mul + 64;
    // (pipelined batches: the tables are built over the whole span, so the worst case -- five times the input a
    // batch of compressible data needs -- would have every batch index most of what is left.  The span is what the
    // batch's RSIs need on average plus the index pass's look-ahead; the pass is told that it sees a piece
    // (aec_gpu_set_index_piece) and stops in front of the RSIs its tables cannot resolve at the end of it -- the next
    // batch's -- instead of walking them serially.)
    bool piece = false;
    if (pipe && hint && s->span_mul == 1 && !s->span_wide) {
        const uint64_t tight = walk_rel / 8 + (max_rsi * hint + 8 * hint) / 8 + 65536;
        if (tight < span && tight < s->d_len) {
            span = tight;
            piece = true;
        }
    }
    const size_t in_bytes = span < s->d_len ? (size_t)span : s->d_len;
    DevBuf &obuf = s->out_sel ? s->d_out2 : s->d_out;
    if (!s->d_off.ensure((max_rsi + 2) * 8) || !obuf.ensure(max_rsi * rsi_bytes + blk_bytes + 64))
        return AEC_FAIL(AEC_MEM_ERROR);
    // (the copy that read this buffer two batches ago must have finished before the decoder writes it again)
    if (s->ev_copied[s->out_sel] && s->copy_pending &&
        hipStreamWaitEvent(s->stream, s->ev_copied[s->out_sel], 0) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    // RSIs of eight segments and more: the index pass also leaves the segment starts, and the decoder takes a lane per
    // segment instead of one per RSI (include/aec_gpu.h: aec_gpu_index_segments_async; without the table -- no memory
    // for it -- a lane per RSI as before)
    uint64_t *d_seg = nullptr;
    if (c.segs_per_rsi >= 8 && s->d_seg.ensure((max_rsi + 2) * c.segs_per_rsi * 8)) d_seg = static_cast<uint64_t *>(s->d_seg.p);

    aec_gpu_dec_result *d_idx = static_cast<aec_gpu_dec_result *>(s->d_res.p), *d_dec = d_idx + 1;
    uint64_t *d_off = static_cast<uint64_t *>(s->d_off.p);
    aec_gpu_set_index_hint(s->ctx, index_hint_of(c, hint));
    if (piece) aec_gpu_set_index_piece(s->ctx, 6 * hint + 8192);
    int rc = d_seg ? aec_gpu_index_segments_async(s->ctx, &s->prm, s->d_in.p, in_bytes, walk_rel, s->walk_blocks, rsi_rel,
                                                  d_off, d_seg, max_rsi, d_idx, s->stream)
                   : aec_gpu_index_resume_async(s->ctx, &s->prm, s->d_in.p, in_bytes, walk_rel, s->walk_blocks, rsi_rel,
                                                d_off, max_rsi, d_idx, s->stream);
    if (rc != RC_OK) return AEC_FAIL(rc);
    rc = d_seg ? aec_gpu_decode_bare_async(s->ctx, &s->prm, s->d_in.p, in_bytes, d_off, d_seg, max_rsi, 0, d_idx,
                                           obuf.p, d_dec, s->stream)
               : aec_gpu_decode_indexed_async(s->ctx, &s->prm, s->d_in.p, in_bytes, d_off, max_rsi, d_idx, obuf.p,
                                              d_dec, s->stream);
    if (rc != RC_OK) return AEC_FAIL(rc);
    // records, the start of the trailing partial RSI, and the first output bytes: one synchronisation
    uint8_t *bounce = s->h_res + 256;
    const uint8_t *d_bytes = static_cast<const uint8_t *>(obuf.p) + skip;
    size_t spec = max_rsi * rsi_bytes - skip;
    if (spec > kBounce) spec = kBounce;
    const auto t_enq = std::chrono::steady_clock::now();                // (AEC_ABI_TRACE: where a batch's time goes)
    if (hipMemcpyAsync(s->h_res, d_idx, 2 * sizeof(aec_gpu_dec_result), hipMemcpyDeviceToHost, s->stream) !=
            hipSuccess ||
        hipMemcpyAsync(s->h_res + 128, d_off + max_rsi, 8, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipMemcpyAsync(bounce, d_bytes, spec, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    if (trace_on()) {
        static thread_local std::chrono::steady_clock::time_point t_prev = t_enq;
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "libaec (MI355X): decode batch: %.3f ms since the batch in front was through, of which %.3f ms "
                "waiting for this one's kernels after the last launch\n",
                std::chrono::duration<double, std::milli>(t_done - t_prev).count(),
                std::chrono::duration<double, std::milli>(t_done - t_enq).count());
        t_prev = t_done;
    }
    const aec_gpu_dec_result idx = reinterpret_cast<aec_gpu_dec_result *>(s->h_res)[0];
    const aec_gpu_dec_result dec = reinterpret_cast<aec_gpu_dec_result *>(s->h_res)[1];
    if (trace_on())
        fprintf(stderr, "libaec (MI355X): decode batch: pipelined %d, room for %llu RSIs, span %zu of %zu resident bytes (from byte %llu), "
                "hint %llu bits per RSI -> %llu RSIs + %llu blocks, walker status %u pad %u\n", (int)pipe, (unsigned long long)max_rsi,
                in_bytes, s->d_len, (unsigned long long)(walk_rel / 8), (unsigned long long)hint, (unsigned long long)idx.n_rsi,
                (unsigned long long)idx.tail_blocks, idx.status, idx.pad);
    uint64_t tail_start = 0;
    memcpy(&tail_start, s->h_res + 128, 8);

    // What is good: all of it, or -- when the decoder met a corrupt coded data set the walker could
    // not see (a second-extension code beyond the table, reference decode.c:589-616) -- the RSIs in
    // front of the first bad one; the reference delivers every sample preceding the error as well.
    uint64_t good_rsi = idx.n_rsi, tail_blocks = idx.tail_blocks;
    bool corrupt = idx.status == DEC_DATA_ERROR;
    if (dec.status != DEC_OK) {
        corrupt = true;
        if (dec.bad_rsi <= good_rsi) {
            // (the decode record's tail_blocks: the lowest failing block of the batch -- the blocks of its RSI in
            // front of it are decoded and good)
            const uint64_t bad_block = dec.tail_blocks;
            good_rsi = dec.bad_rsi;
            tail_blocks = (bad_block != ~0ull && bad_block / c.rsi == good_rsi) ? bad_block % c.rsi : 0;
        }
    }
    // samples released from the coded data set the input ends in (reference decode.c:423-460)
    const uint32_t part = (!corrupt && idx.pad == 1) ? (dec.pad & 0x7FFFFFFFu) : 0u;   // (bit 31: slow path ran)
    const uint64_t blocks = good_rsi * c.rsi + tail_blocks;
    const size_t total = (size_t)blocks * blk_bytes + (size_t)part * c.bytes;
    if (total > skip) {
        const size_t fresh = total - skip;
        size_t direct = fresh < strm->avail_out ? fresh : strm->avail_out;
        direct -= direct % c.bytes;
        if (direct) {
            const size_t from_bounce = direct < spec ? direct : spec;
            memcpy(strm->next_out, bounce, from_bounce);
            if (direct > from_bounce) {
                const size_t rest = direct - from_bounce;
                // a large rest goes on the side stream: the next batch's kernels (the other output buffer) run beside
                // it, and the call waits for it before it returns (copies_done)
                bool async = pipe && rest >= kAsyncMin;
                if (async && !s->copy_stream) async = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) == hipSuccess;
                if (async && !s->ev_copied[s->out_sel])
                    async = hipEventCreateWithFlags(&s->ev_copied[s->out_sel], hipEventDisableTiming) == hipSuccess;
                if (async) {
                    if (hipMemcpyAsync(strm->next_out + from_bounce, d_bytes + from_bounce, rest, hipMemcpyDeviceToHost,
                                       s->copy_stream) != hipSuccess ||
                        hipEventRecord(s->ev_copied[s->out_sel], s->copy_stream) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                    s->copy_pending = true;
                    s->out_sel ^= 1u;
                } else {
                    (void)hipGetLastError();
                    if (hipMemcpy(strm->next_out + from_bounce, d_bytes + from_bounce, rest, hipMemcpyDeviceToHost) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                }
            }
            strm->next_out += direct;
            strm->avail_out -= direct;
        }
        if (fresh > direct) {
            const size_t rest = fresh - direct, at = s->outq.size();
            s->outq.resize(at + rest);
            size_t done = 0;
            if (direct < spec) {
                done = spec - direct < rest ? spec - direct : rest;
                memcpy(s->outq.data() + at, bounce + direct, done);
            }
            if (rest > done &&
                hipMemcpy(s->outq.data() + at + done, d_bytes + direct + done, rest - done, hipMemcpyDeviceToHost) !=
                    hipSuccess)
                return AEC_FAIL(AEC_MEM_ERROR);
        }
    }
    // A coded data set that cannot be accepted -- a zero run overrunning its RSI (found by the walker), a
    // second-extension code beyond the table (found by the decoder) -- BEHIND everything this call was asked for is
    // not this call's error: the reference stops when the output is full (decode.c:797-831) and only meets it
    // when it is asked for more, as the next call here will.  The walk then resumes at the coded data set the walker
    // stopped at, or at the start of the RSI the decoder gave up.
    uint64_t res_rsi = idx.n_rsi, res_tail = idx.tail_blocks, res_end = idx.end_bit;
    bool more_behind = false;
    // (a call that offers NO room defers what the walker found -- the reference's m_zero_block only refuses a run when
    // avail_out holds it, decode.c:542-544 -- but not a second-extension code beyond the table, which the reference
    // reports whatever the room, decode.c:589-616)
    if (corrupt && total >= skip + want_out && (want_out || dec.status == DEC_OK)) {
        corrupt = false;
        more_behind = true;                                 // (the next call that offers room meets the error)
        if (dec.status != DEC_OK) {
            uint64_t off = 0;
            if (hipMemcpy(&off, d_off + good_rsi, 8, hipMemcpyDeviceToHost) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
            res_rsi = good_rsi;
            res_tail = 0;                                   // (the walker starts the RSI again; its good blocks are `delivered`)
            res_end = off;
        }
    }
    if (corrupt) {
        if (trace_on())
            fprintf(stderr, "libaec (MI355X): AEC_DATA_ERROR: walker status %u after %llu RSIs + %llu blocks (bit %llu), "
                    "decoder status %u at RSI %llu\n", idx.status, (unsigned long long)idx.n_rsi,
                    (unsigned long long)idx.tail_blocks, (unsigned long long)idx.end_bit, dec.status,
                    (unsigned long long)dec.bad_rsi);
        return AEC_DATA_ERROR;
    }
    if (res_rsi) s->rsi_bits_seen = (res_end - rsi_rel) / (res_rsi + (res_tail ? 1 : 0));

    // advance: the walker resumes behind the last complete coded data set
    s->walk_bit = base_bits + res_end;
    s->walk_blocks = (uint32_t)res_tail;
    if (res_tail) {
        s->rsi_start_bit = base_bits + tail_start;
        s->delivered = res_tail * c.bs + part;
    } else {
        s->rsi_start_bit = s->walk_bit;
        s->delivered = (more_behind && dec.status != DEC_OK) ? tail_blocks * c.bs : part;
    }
    s->walked_len = in_bytes;
    // (the walker ran out of input inside the span although more is resident: first the worst case instead of the
    // tight span of the pipelined batches, then wider and wider)
    if (idx.pad == 1 && in_bytes < s->d_len) {
        if (piece && !s->span_wide) s->span_wide = true;
        else s->span_mul = s->span_mul < (1u << 20) ? s->span_mul * 4 : s->span_mul;
    } else {
        s->span_mul = 1;
    }
    // stopped at the bound with input left: the caller's next call (or this one, if it still has
    // room) goes on from here
    s->more = idx.n_rsi >= max_rsi || in_bytes < s->d_len || more_behind;

    // drop the consumed front of the resident stream once it is the larger part (the copy must not overlap)
    const uint64_t keep_from = (s->rsi_start_bit / 8 - s->base) & ~(uint64_t)15;
    const size_t rem = s->d_len - (size_t)keep_from;
    if (keep_from >= rem && keep_from >= 4096) {
        if (rem && hipMemcpyAsync(s->d_in.p, static_cast<uint8_t *>(s->d_in.p) + keep_from, rem,
                                  hipMemcpyDeviceToDevice, s->stream) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        s->base += keep_from;
        s->d_len = rem;
        s->walked_len = s->walked_len > keep_from ? s->walked_len - (size_t)keep_from : 0;
    }
    return AEC_OK;
}

int encode_call(struct aec_stream *strm, int flush)
{
    internal_state *s = strm->state;
    const size_t bytes = s->cfg.bytes;
    const size_t rsi_bytes = (size_t)s->cfg.rsi * s->cfg.bs * bytes;
    s->flush = flush;
    strm->total_in += strm->avail_in;     // reference encode.c:919-920
    strm->total_out += strm->avail_out;
    const bool brought = strm->avail_in >= bytes;
    int rc = AEC_OK;

    for (;;) {
        drain(strm, s, 1);
        if (!s->outq.empty()) break;      // output full
        if (s->finished) {
            s->flushed = true;            // reference encode.c:689-694
            break;
        }
        // only whole samples are ever consumed (reference encode.c:673-674)
        const size_t take = strm->avail_in - strm->avail_in % bytes;
        size_t staged = s->stage.size() - s->stage_pos;
        if (staged == 0 && take >= rsi_bytes && (take >= kEncDirectMin || flush == AEC_FLUSH)) {
            // whole RSIs offered and nothing staged: code them from the caller's buffer
            const size_t direct = take / rsi_bytes * rsi_bytes;
            rc = encode_batch(s, strm->next_in, direct, strm);
            strm->next_in += direct;
            strm->avail_in -= direct;
            if (rc != AEC_OK) break;
            continue;
        }
        if (take) {
            if (s->stage_pos && s->stage_pos == s->stage.size()) {
                s->stage.clear();
                s->stage_pos = 0;
            }
            s->stage.insert(s->stage.end(), strm->next_in, strm->next_in + take);
            strm->next_in += take;
            strm->avail_in -= take;
            staged += take;
        }
        const size_t whole = staged / rsi_bytes * rsi_bytes;
        // A launch is worth it for a batch of some size, is due on AEC_FLUSH, and is what a caller asks
        // for who comes back without new input while whole RSIs wait (the reference would have coded
        // them by now: tests/check_aec.c encode_decode_small collects its output that way).
        if (whole && (flush == AEC_FLUSH || whole >= kEncBatchBytes || !brought)) {
            rc = encode_batch(s, s->stage.data() + s->stage_pos, whole, strm);
            s->stage_pos += whole;
            if (s->stage_pos == s->stage.size()) {
                s->stage.clear();
                s->stage_pos = 0;
            } else if (s->stage_pos >= s->stage.size() - s->stage_pos) {
                s->stage.erase(s->stage.begin(), s->stage.begin() + (ptrdiff_t)s->stage_pos);
                s->stage_pos = 0;
            }
            if (rc != AEC_OK) break;
            continue;
        }
        if (flush == AEC_FLUSH) {
            // last, short RSI (reference encode.c:676-684), then the final byte (686-695)
            rc = encode_batch(s, s->stage.data() + s->stage_pos, staged, strm);
            s->stage.clear();
            s->stage_pos = 0;
            if (rc != AEC_OK) break;
            if (s->part_bits || !s->any_bits) s->outq.push_back(s->part_byte);
            s->part_bits = 0;
            s->part_byte = 0;
            s->finished = true;
            continue;
        }
        break;                            // need more input
    }
    strm->total_in -= strm->avail_in;     // reference encode.c:933-934
    strm->total_out -= strm->avail_out;
    return rc;
}

int decode_call(struct aec_stream *strm, int flush)
{
    internal_state *s = strm->state;
    const size_t bytes = s->cfg.bytes;
    strm->total_in += strm->avail_in;     // reference decode.c:811-812
    strm->total_out += strm->avail_out;
    int rc = s->sticky_error;
    const size_t brought = strm->avail_in;

    while (rc == AEC_OK) {
        drain(strm, s, bytes);
        if (!s->outq.empty()) break;      // output full (or less than one sample of room)
        // Accept input while the undecoded backlog on the device is moderate; large pieces go straight
        // to the device, trickles are collected on the host first.
        // (a caller that offers more room than that may bring as much input: the stream of a one-shot decode goes up
        // whole and is indexed in one pass -- in pieces of 64 MiB a 180 MB stream of the reference's sample shape paid
        // the fixed phases of the trunk index three times, 71 ms instead of 48 for 256 MiB)
        // (as much input as that output can take at most -- incompressible data is longer coded than decoded, and a
        // stream cut at the size of its output had its last few per cent indexed and decoded as a second batch, with
        // an average coded RSI for the first that was too low to tell what kind of stream it is)
        const size_t rsi_out = (size_t)s->cfg.rsi * s->cfg.bs * bytes;
        const size_t in_for_out = (size_t)((strm->avail_out / rsi_out + 2) * worst_rsi_bytes(s->cfg));
        const size_t backlog_max = strm->avail_out >= kBacklogMax ? (in_for_out > strm->avail_out ? in_for_out : strm->avail_out)
                                                                  : kBacklogMax;
        if (strm->avail_in && s->d_len - (size_t)((s->walk_bit / 8) - s->base) < backlog_max) {
            size_t n = strm->avail_in < backlog_max ? strm->avail_in : backlog_max;
            if (n >= kDecDirectMin && s->stage.empty()) {
                rc = upload(s, strm->next_in, n);
                // (the staging buffer for a one-shot caller's whole stream could not be allocated: in pieces of the
                // ordinary backlog, as before round 4 -- only then AEC_MEM_ERROR.  A stream that goes up whole and then
                // finds no memory for its output or its index workspace is NOT retried here: decode_run reports it, or
                // the index pass falls back to a smaller workspace / the serial walk: aec_gpu.hip index_common)
                if (rc == AEC_MEM_ERROR && n > kBacklogMax) {
                    (void)hipGetLastError();
                    n = kBacklogMax;
                    rc = upload(s, strm->next_in, n);
                }
                if (rc != AEC_OK) break;
            } else {
                s->stage.insert(s->stage.end(), strm->next_in, strm->next_in + n);
            }
            strm->next_in += n;
            strm->avail_in -= n;
        }
        const bool pending = !s->stage.empty() || s->d_len > s->walked_len || s->more;
        if (!pending) break;
        // When to run a batch: always -- a caller may take "output not full" to mean that everything
        // the input allows has come out -- except for a caller that trickles input in a few bytes per
        // call (at most kDecTinyCall new bytes, fewer than kDecTrickle waiting, AEC_NO_FLUSH, not the
        // first call): those bytes are collected until a call brings nothing new, which is how such
        // callers ask for the rest (reference src/aec.c:191-221, tests/check_aec.c:138-166), so that a
        // byte-at-a-time caller costs a vector append per call instead of a launch.
        const bool trickle = brought && brought <= kDecTinyCall && flush != AEC_FLUSH && s->launched && !s->more &&
                             s->d_len <= s->walked_len && s->stage.size() < kDecTrickle;
        if (trickle) break;
        const size_t out_before = strm->avail_out, q_before = s->outq.size();
        const uint64_t walk_before = s->walk_bit, span_before = s->span_mul;
        const bool wide_before = s->span_wide;
        rc = decode_run(s, strm);
        if (rc == AEC_DATA_ERROR) s->sticky_error = rc;
        if (rc != AEC_OK) break;
        const bool progressed = out_before != strm->avail_out || q_before != s->outq.size() ||
                                walk_before != s->walk_bit || s->span_mul > span_before ||
                                s->span_wide != wide_before;                               // (a wider span is tried at once)
        if (!progressed) break;           // what is here needs more input before anything else comes out
    }
    // (copies to the caller's buffer that run on the side stream: the buffer is the caller's again on return)
    if (s->copy_pending) {
        s->copy_pending = false;
        if (hipStreamSynchronize(s->copy_stream) != hipSuccess && rc == AEC_OK) rc = AEC_FAIL(AEC_MEM_ERROR);
    }
    if (rc == AEC_DATA_ERROR) drain(strm, s, bytes);   // the samples in front of the error are delivered
    if (rc != AEC_OK) return rc;          // reference decode.c:818-819 (totals left as they are)
    if (strm->avail_out > 0 && strm->avail_out < bytes) return AEC_FAIL(AEC_MEM_ERROR);   // decode.c:821-823
    strm->total_in -= strm->avail_in;     // reference decode.c:827-828
    strm->total_out -= strm->avail_out;
    return AEC_OK;
}

// ---- many independent streams per call (include/libaec.h: aec_buffer_*_batch) -------------------------
struct BatchKit {
    Kit k;
    bool ok = false;
    explicit BatchKit(int device)
    {
        if (take_kit(device, &k)) { ok = true; return; }
        k.device = device;
        ok = aec_gpu_create(&k.ctx) == RC_OK && hipStreamCreate(&k.stream) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void **>(&k.h_res), 256 + kBounce, hipHostMallocDefault) == hipSuccess &&
             k.d_res.ensure(256);
    }
    ~BatchKit()
    {
        if (ok && k.ctx && k.stream && k.h_res && k.d_res.p) park_kit(k);
        else destroy_kit(k);
    }
};

inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

// pinned staging buffer of a kit, at least n bytes (false: not to be had -- the callers copy chunk by chunk then)
bool stage_ensure(Kit &k, size_t n)
{
    if (n <= k.h_stage_cap) return true;
    if (k.h_stage) (void)hipHostFree(k.h_stage);
    k.h_stage = nullptr;
    k.h_stage_cap = 0;
    if (hipHostMalloc(reinterpret_cast<void **>(&k.h_stage), n, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    k.h_stage_cap = n;
    return true;
}

// many host-to-host copies (caller's buffers <-> pinned staging), on a few threads when there is enough to copy
struct CopyJob {
    void *dst;
    const void *src;
    size_t n;
};
thread_local bool t_in_part = false;          // this thread is one of run_parts' workers: no threads of its own

// Waiting for a kit's stream inside the batch paths.  With several host threads each waiting on a stream of its own,
// hipStreamSynchronize was measured to return up to 9 ms late now and then (2.4 ms batches taking 10); polling the
// stream does not.  The poll yields for the first few dozen microseconds -- a small batch is through by then -- and then
// sleeps between two looks (round 6: a part thread burnt a core for the whole of its batch; the timer's slack, ~60 us,
// is a few per cent of the milliseconds such a batch takes).
hipError_t batch_sync(hipStream_t st)
{
    if (!t_in_part) return hipStreamSynchronize(st);
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (spins < 256u) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(25));
    }
}

void copy_all(const std::vector<CopyJob> &jobs)
{
    size_t total = 0;
    for (const CopyJob &j : jobs) total += j.n;
    const unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = total >= ((size_t)8 << 20) && !t_in_part ? (hw >= 8 ? 4u : (hw >= 2 ? 2u : 1u)) : 1u;
    if (nt > jobs.size()) nt = (unsigned)jobs.size();
    auto work = [&](unsigned t) {
        for (size_t i = t; i < jobs.size(); i += nt)
            if (jobs[i].n) memcpy(jobs[i].dst, jobs[i].src, jobs[i].n);
    };
    if (nt <= 1) {
        work(0);
        return;
    }
    // (the pool's threads, kept between calls -- round 5; until then a thread per share was created and joined per call)
    WorkerPool::run(nt, [&](size_t t) { work((unsigned)t); });
}

// A large batch as several parts side by side: every part is a batch of its own on its own kit (HIP stream,
// context, buffers), driven by its own host thread -- the uploads of one part run beside the kernels and the
// downloads of the others, and the launch sequences of many small chunks (5 launches per coded chunk, ~3 us of
// host time each) are issued from several threads.  part(lo, hi) handles chunks [lo, hi) and returns what the
// batch call would return for them; the call returns the hard error of a part if there is one, else the
// status of the last chunk that has one (as one batch does).
template <class Part>
int run_parts(size_t n, size_t total_bytes, Part part)
{
    const unsigned hw = std::thread::hardware_concurrency();
    size_t parts = total_bytes / ((size_t)4 << 20);
    // (at most 4: with 8, every part's synchronisation spinning on a core, the same batch took 9 ms instead of 2.4
    // on the 16 cores of the test machine)
    if (parts > 4) parts = 4;
    if (hw && parts > hw / 4) parts = hw / 4;
    if (parts > n / 4) parts = n / 4;
    if (parts <= 1 || t_in_part) return part((size_t)0, n);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    std::vector<int> rcs(parts, AEC_OK);
    // (the threads are kept between calls: aec_pool.h)
    WorkerPool::run(parts, [&](size_t t) {
        (void)hipSetDevice(device);
        const bool outer = !t_in_part;
        t_in_part = true;
        try { rcs[t] = part(n * t / parts, n * (t + 1) / parts); }
        catch (const std::bad_alloc &) { rcs[t] = AEC_MEM_ERROR; }
        catch (...) { rcs[t] = AEC_MEM_ERROR; }
        if (outer) t_in_part = false;
    });
    int rc = AEC_OK;
    for (size_t t = 0; t < parts; t++)
        if (rcs[t] != AEC_OK) rc = rcs[t];
    for (size_t t = 0; t < parts; t++)
        if (rcs[t] == AEC_MEM_ERROR || rcs[t] == AEC_CONF_ERROR) rc = rcs[t];
    return rc;
}

int decode_batch(const struct aec_stream *prm, size_t n, const void *const *src, const size_t *src_len,
                 void *const *dst, size_t *dst_len, int *status)
{
    aec_gpu_params gp{prm->bits_per_sample, prm->block_size, prm->rsi, prm->flags};
    Cfg c;
    int rc = make_cfg(gp.bits_per_sample, gp.block_size, gp.rsi, gp.flags, 0, false, &c);
    if (rc != RC_OK) return AEC_FAIL(rc);
    if (n == 0) return AEC_OK;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    BatchKit bk(device);
    if (!bk.ok) return AEC_FAIL(AEC_MEM_ERROR);
    Kit &k = bk.k;
    aec_gpu_set_index_hint(k.ctx, 0);      // (a kit from the pool: whatever its last stream measured is not this batch's)
    const size_t blk_bytes = (size_t)c.bs * c.bytes, rsi_bytes = (size_t)c.rsi * blk_bytes;
    // geometry: every stream gets room for the RSIs of the largest one
    uint64_t rpc = 1;
    std::vector<uint64_t> off(n + 1);
    size_t total_in = 0;
    for (size_t i = 0; i < n; i++) {
        off[i] = total_in;
        total_in += up16(src_len[i]) + 16;
        const uint64_t r = (dst_len[i] + rsi_bytes - 1) / rsi_bytes;
        if (r > rpc) rpc = r;
    }
    off[n] = total_in;
    // the index walker takes [off[i], off[i+1]) as stream i: the padding behind a stream is zeroed (zero
    // bits never complete a coded data set), the streams go up straight from the caller's buffers
    // (chunk offsets twice: the n + 1 absolute ones, and -- for the table path, which takes the batch in groups --
    // relative to the group a chunk belongs to, every group with a closing entry of its own: n + G entries for G <= n
    // groups, so the region holds 3 n + 4)
    const size_t o_choff = up16((size_t)n * rpc * 8), o_res = o_choff + up16((3 * n + 4) * 8),
                 o_one = o_res + up16(n * 40);
    if (!k.d_in.ensure(total_in + 32) || !k.d_out.ensure((size_t)n * rpc * rsi_bytes + 64) ||
        !k.d_off.ensure(o_one + 64))
        return AEC_FAIL(AEC_MEM_ERROR);
    uint8_t *meta = static_cast<uint8_t *>(k.d_off.p);
    if (n >= 4 && total_in <= kStagePiece && stage_ensure(k, total_in)) {
        // the streams and the zero padding behind each of them, assembled in pinned memory: ONE transfer
        std::vector<CopyJob> jobs(n);
        for (size_t i = 0; i < n; i++) {
            jobs[i] = CopyJob{k.h_stage + off[i], src[i], src_len[i]};
            memset(k.h_stage + off[i] + src_len[i], 0, (size_t)(off[i + 1] - off[i]) - src_len[i]);
        }
        copy_all(jobs);
        if (hipMemcpyAsync(k.d_in.p, k.h_stage, total_in, hipMemcpyHostToDevice, k.stream) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
    } else {
        if (hipMemsetAsync(k.d_in.p, 0, total_in, k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
        for (size_t i = 0; i < n; i++)
            if (src_len[i] && hipMemcpyAsync(static_cast<uint8_t *>(k.d_in.p) + off[i], src[i], src_len[i],
                                             hipMemcpyHostToDevice, k.stream) != hipSuccess)
                return AEC_FAIL(AEC_MEM_ERROR);
    }
    if (hipMemcpyAsync(meta + o_choff, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, k.stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    // Small chunks: one wavefront walks each stream, all streams at once (aec_gpu_decode_batch_async).
    // Large chunks would each keep ONE wavefront busy for tens of milliseconds that way; they go through
    // the speculative index one after the other instead (all CUs per chunk), still without returning to
    // the host in between.
    aec_gpu_dec_result *d_results = reinterpret_cast<aec_gpu_dec_result *>(meta + o_res);
    aec_gpu_dec_result *d_one = reinterpret_cast<aec_gpu_dec_result *>(meta + o_one);
    const bool large = total_in / n >= ((size_t)32 << 10) || n < 64;
    // Large low-entropy chunks: groups of about 12 MiB of streams, each group ONE table launch + one wavefront
    // per stream + one decode launch (aec_gpu_decode_batch_async)
    constexpr size_t kGroupBytes = (size_t)12 << 20;
    bool grouped = false;
    if (large && n >= 2) {
        size_t probe_n = n, probe_bytes = total_in;
        if (total_in > kGroupBytes) {
            probe_n = (size_t)((uint64_t)n * kGroupBytes / total_in);
            if (probe_n < 1) probe_n = 1;
            probe_bytes = total_in / n * probe_n;
        }
        grouped = aec_gpu_batch_uses_tables(k.ctx, &gp, probe_bytes, probe_n, rpc) != 0;
    }
    if (grouped) {
        std::vector<uint64_t> rel;
        std::vector<size_t> first;                      // first chunk of every group, index of its offsets in rel
        std::vector<size_t> at;
        for (size_t i = 0; i < n;) {
            size_t j = i;
            while (j < n && (j == i || off[j + 1] - off[i] <= kGroupBytes)) j++;
            first.push_back(i);
            at.push_back(rel.size());
            for (size_t q = i; q <= j; q++) rel.push_back(off[q] - off[i]);
            i = j;
        }
        first.push_back(n);
        if (n + 1 + rel.size() > 3 * n + 4) return AEC_FAIL(AEC_MEM_ERROR);          // (cannot happen: rel holds n + G <= 2 n)
        uint64_t *d_rel = reinterpret_cast<uint64_t *>(meta + o_choff) + (n + 1);
        if (hipMemcpyAsync(d_rel, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, k.stream) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        for (size_t gi = 0; gi + 1 < first.size() && rc == RC_OK; gi++) {
            const size_t i0 = first[gi], i1 = first[gi + 1];
            rc = aec_gpu_decode_batch_async(k.ctx, &gp, static_cast<const uint8_t *>(k.d_in.p) + off[i0], off[i1] - off[i0],
                                            d_rel + at[gi], i1 - i0, rpc, reinterpret_cast<uint64_t *>(meta) + i0 * rpc,
                                            static_cast<uint8_t *>(k.d_out.p) + i0 * rpc * rsi_bytes, d_results + i0, d_one,
                                            k.stream);
        }
        if (rc != RC_OK) return AEC_FAIL(rc);
        // (decoder-side errors are in the chunks' own records; the overall record belongs to the last group only)
        if (hipMemsetAsync(d_one, 0, sizeof(aec_gpu_dec_result), k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    } else if (!large) {
        rc = aec_gpu_decode_batch_async(k.ctx, &gp, k.d_in.p, total_in, reinterpret_cast<uint64_t *>(meta + o_choff), n,
                                        rpc, reinterpret_cast<uint64_t *>(meta), k.d_out.p, d_results, d_one, k.stream);
        if (rc != RC_OK) return AEC_FAIL(rc);
    } else {
        // (per chunk: its own decode record behind the index records; the overall record is folded on the host)
        // (the per-chunk decode records live behind the stream's own record in d_res, which stays with the kit)
        if (!k.d_res.ensure(256 + n * sizeof(aec_gpu_dec_result) + 64)) return AEC_FAIL(AEC_MEM_ERROR);
        aec_gpu_dec_result *d_dec = reinterpret_cast<aec_gpu_dec_result *>(static_cast<uint8_t *>(k.d_res.p) + 256);
        for (size_t i = 0; i < n && rc == RC_OK; i++) {
            uint64_t *offs = reinterpret_cast<uint64_t *>(meta) + i * rpc;
            const uint8_t *in_i = static_cast<const uint8_t *>(k.d_in.p) + off[i];
            aec_gpu_set_index_hint(k.ctx, rpc ? (uint64_t)src_len[i] * 8 / rpc : 0);
            rc = aec_gpu_index_async(k.ctx, &gp, in_i, src_len[i], 0, offs, rpc, d_results + i, k.stream);
            if (rc == RC_OK)
                rc = aec_gpu_decode_indexed_async(k.ctx, &gp, in_i, src_len[i], offs, rpc, d_results + i,
                                                  static_cast<uint8_t *>(k.d_out.p) + i * rpc * rsi_bytes, d_dec + i,
                                                  k.stream);
        }
        std::vector<aec_gpu_dec_result> dec(n);
        if (rc != RC_OK || hipMemcpyAsync(dec.data(), d_dec, n * sizeof(aec_gpu_dec_result), hipMemcpyDeviceToHost,
                                          k.stream) != hipSuccess ||
            hipMemsetAsync(d_one, 0, sizeof(aec_gpu_dec_result), k.stream) != hipSuccess ||
            batch_sync(k.stream) != hipSuccess) {
            return AEC_FAIL(rc != RC_OK ? rc : AEC_MEM_ERROR);
        }
        // fold the per-chunk decode status into the per-chunk index records on the host below
        std::vector<aec_gpu_dec_result> idx(n);
        if (hipMemcpy(idx.data(), d_results, n * sizeof(aec_gpu_dec_result), hipMemcpyDeviceToHost) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        for (size_t i = 0; i < n; i++)
            if (dec[i].status != DEC_OK) idx[i].status = DEC_DATA_ERROR;
        if (hipMemcpy(d_results, idx.data(), n * sizeof(aec_gpu_dec_result), hipMemcpyHostToDevice) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
    }
    std::vector<aec_gpu_dec_result> res(n + 1);
    if (hipMemcpyAsync(res.data(), meta + o_res, n * sizeof(aec_gpu_dec_result), hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
        hipMemcpyAsync(&res[n], meta + o_one, sizeof(aec_gpu_dec_result), hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
        batch_sync(k.stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    int worst = AEC_OK;
    // the outputs: through pinned staging in pieces of whole slots (one transfer per piece, then copies to the
    // callers' buffers on a few threads), or chunk by chunk where there is no staging to be had
    // (pieces of about 4 MiB through the two halves of the staging buffer: the transfer of piece p + 1 runs beside the
    // host's copies of piece p -- one transfer of a whole part and then its copies were the two largest items of a
    // batch of 64 x 1 MiB behind the index kernels)
    const size_t slot_out = (size_t)rpc * rsi_bytes;
    constexpr size_t kOutPiece = (size_t)4 << 20;
    const size_t per_piece = !slot_out ? 0 : (slot_out <= kOutPiece ? kOutPiece / slot_out : (slot_out <= kStagePiece / 2 ? 1 : 0));
    const size_t npieces = per_piece ? (n + per_piece - 1) / per_piece : 0;
    const size_t half_bytes = per_piece * slot_out;
    bool staged = n >= 4 && per_piece && stage_ensure(k, (npieces > 1 ? 2 : 1) * (n < per_piece ? n : per_piece) * slot_out);
    for (int e = 0; e < 2 && staged; e++)
        if (!k.ev_copied[e]) staged = hipEventCreateWithFlags(&k.ev_copied[e], hipEventDisableTiming) == hipSuccess;
    auto fetch = [&](size_t piece) -> bool {         // the transfer of a piece into half (piece & 1)
        const size_t first = piece * per_piece, cnt = n - first < per_piece ? n - first : per_piece;
        return hipMemcpyAsync(k.h_stage + (piece & 1) * half_bytes, static_cast<uint8_t *>(k.d_out.p) + first * slot_out,
                              cnt * slot_out, hipMemcpyDeviceToHost, k.stream) == hipSuccess &&
               hipEventRecord(k.ev_copied[piece & 1], k.stream) == hipSuccess;
    };
    auto landed = [&](size_t piece) -> bool {
        if (!t_in_part) return hipEventSynchronize(k.ev_copied[piece & 1]) == hipSuccess;
        for (unsigned spins = 0;; spins++) {         // (polling, as batch_sync does)
            const hipError_t e = hipEventQuery(k.ev_copied[piece & 1]);
            if (e != hipErrorNotReady) return e == hipSuccess;
            if (spins < 256u) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(25));
        }
    };
    if (staged && !fetch(0)) return AEC_FAIL(AEC_MEM_ERROR);
    std::vector<CopyJob> jobs;
    for (size_t i = 0; i < n; i++) {
        const uint64_t blocks = res[i].n_rsi * c.rsi + res[i].tail_blocks;
        size_t produced = (size_t)blocks * blk_bytes;
        if (produced > dst_len[i]) produced = dst_len[i] - dst_len[i] % c.bytes;
        int st = res[i].status == DEC_DATA_ERROR ? AEC_DATA_ERROR : AEC_OK;
        if (!grouped && res[n].status != DEC_OK && res[n].bad_rsi / rpc == i) st = AEC_DATA_ERROR;
        if (staged) {
            const size_t piece = i / per_piece, first = piece * per_piece;
            if (i == first) {                            // a new piece: the next one sets out, this one has to be here
                if ((piece + 1 < npieces && !fetch(piece + 1)) || !landed(piece)) return AEC_FAIL(AEC_MEM_ERROR);
                jobs.clear();
            }
            jobs.push_back(CopyJob{dst[i], k.h_stage + (piece & 1) * half_bytes + (i - first) * slot_out, produced});
            if (i + 1 == n || (i + 1) % per_piece == 0) copy_all(jobs);
        } else if (produced && hipMemcpyAsync(dst[i], static_cast<uint8_t *>(k.d_out.p) + i * slot_out, produced,
                                              hipMemcpyDeviceToHost, k.stream) != hipSuccess) {
            return AEC_FAIL(AEC_MEM_ERROR);
        }
        dst_len[i] = produced;
        if (status) status[i] = st;
        if (st != AEC_OK) worst = st;
    }
    if (batch_sync(k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    return worst;
}

int encode_batch_host(const struct aec_stream *prm, size_t n, const void *const *src, const size_t *src_len,
                      void *const *dst, size_t *dst_len, int *status)
{
    aec_gpu_params gp{prm->bits_per_sample, prm->block_size, prm->rsi, prm->flags};
    Cfg c;
    int rc = make_cfg(gp.bits_per_sample, gp.block_size, gp.rsi, gp.flags, 0, true, &c);
    if (rc != RC_OK) return AEC_FAIL(rc);
    if (n == 0) return AEC_OK;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    BatchKit bk(device);
    if (!bk.ok) return AEC_FAIL(AEC_MEM_ERROR);
    Kit &k = bk.k;
    std::vector<uint64_t> off(n + 1);
    size_t total_in = 0, largest = 0;
    for (size_t i = 0; i < n; i++) {
        off[i] = total_in;
        const size_t whole = src_len[i] - src_len[i] % c.bytes;          // only whole samples are coded
        total_in += up16(whole) + 16;
        if (whole > largest) largest = whole;
    }
    off[n] = total_in;
    // Equal chunks of whole RSIs (what HDF5 hands the SZIP filter): ONE launch set for all of them, the streams
    // back to back on the device, one transfer each way (aec_gpu_encode_uniform_batch_async)
    {
        bool equal = n >= 2 && src_len[0] && src_len[0] % c.bytes == 0;
        for (size_t i = 1; i < n && equal; i++) equal = src_len[i] == src_len[0];
        const size_t len = src_len[0];
        if (equal && aec_gpu_uniform_batch_ok(&gp, len, n)) {
            const size_t bound = up16(aec_gpu_encode_bound(&gp, len)), cap = n * bound;
            const size_t o_rec = up16(n * sizeof(aec_gpu_batch_chunk));
            if (!k.d_in.ensure(n * len + 32) || !k.d_out.ensure(cap) || !k.d_off.ensure(o_rec + 64))
                return AEC_FAIL(AEC_MEM_ERROR);
            aec_gpu_batch_chunk *d_chunks = static_cast<aec_gpu_batch_chunk *>(k.d_off.p);
            aec_gpu_enc_result *d_one = reinterpret_cast<aec_gpu_enc_result *>(static_cast<uint8_t *>(k.d_off.p) + o_rec);
            // up: through pinned staging in pieces (host copies on a few threads, one transfer per piece)
            const size_t per_up = len <= kStagePiece ? kStagePiece / len : 0;
            // (also large chunks: copies from and to pageable memory issued by several threads at once were measured
            // erratic -- 2.4 or 9 ms for the same 64 MiB -- while pinned transfers plus plain memcpy are steady)
            const bool stage_up = per_up && stage_ensure(k, (n < per_up ? n : per_up) * len);
            for (size_t i = 0; i < n;) {
                if (stage_up) {
                    const size_t cnt = n - i < per_up ? n - i : per_up;
                    if (i && batch_sync(k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
                    std::vector<CopyJob> jobs;
                    for (size_t q = 0; q < cnt; q++) jobs.push_back(CopyJob{k.h_stage + q * len, src[i + q], len});
                    copy_all(jobs);
                    if (hipMemcpyAsync(static_cast<uint8_t *>(k.d_in.p) + i * len, k.h_stage, cnt * len, hipMemcpyHostToDevice,
                                       k.stream) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                    i += cnt;
                } else {
                    if (hipMemcpyAsync(static_cast<uint8_t *>(k.d_in.p) + i * len, src[i], len, hipMemcpyHostToDevice,
                                       k.stream) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                    i++;
                }
            }
            rc = aec_gpu_encode_uniform_batch_async(k.ctx, &gp, k.d_in.p, len, n, k.d_out.p, cap, d_chunks, d_one, k.stream);
            if (rc != RC_OK) return AEC_FAIL(rc);

            std::vector<aec_gpu_batch_chunk> rec(n);
            aec_gpu_enc_result one{};
            if (hipMemcpyAsync(rec.data(), d_chunks, n * sizeof(aec_gpu_batch_chunk), hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
                hipMemcpyAsync(&one, d_one, sizeof(one), hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
                batch_sync(k.stream) != hipSuccess)
                return AEC_FAIL(AEC_MEM_ERROR);
            if (one.overflow) return AEC_FAIL(AEC_MEM_ERROR);                     // (cannot happen: cap is the sum of the bounds)

            // down: the packed streams in pieces of whole streams through the staging buffer
            int worst = AEC_OK;
            const size_t total = (size_t)(one.total_bits / 8);
            const bool stage_down = stage_ensure(k, total < kStagePiece ? (total ? total : 16) : kStagePiece);
            for (size_t i = 0; i < n;) {
                const size_t lo = (size_t)(rec[i].base_bits / 8);
                size_t j = i, hi = lo;
                while (j < n && (size_t)(rec[j].base_bits / 8) + (size_t)((rec[j].bits + 7) / 8) - lo <= k.h_stage_cap) {
                    hi = (size_t)(rec[j].base_bits / 8) + (size_t)((rec[j].bits + 7) / 8);
                    j++;
                }
                const bool piece = stage_down && j > i;
                if (piece) {
                    if (hipMemcpyAsync(k.h_stage, static_cast<uint8_t *>(k.d_out.p) + lo, hi - lo, hipMemcpyDeviceToHost,
                                       k.stream) != hipSuccess ||
                        batch_sync(k.stream) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                } else {
                    j = i + 1;
                }
                std::vector<CopyJob> jobs;
                for (size_t q = i; q < j; q++) {
                    size_t bytes = (size_t)((rec[q].bits + 7) / 8);
                    int st = AEC_OK;
                    if (bytes > dst_len[q]) { st = AEC_STREAM_ERROR; bytes = dst_len[q]; }     // as aec_buffer_encode: a prefix
                    const size_t at = (size_t)(rec[q].base_bits / 8);
                    if (piece) jobs.push_back(CopyJob{dst[q], k.h_stage + (at - lo), bytes});
                    else if (bytes && hipMemcpyAsync(dst[q], static_cast<uint8_t *>(k.d_out.p) + at, bytes, hipMemcpyDeviceToHost,
                                                     k.stream) != hipSuccess)
                        return AEC_FAIL(AEC_MEM_ERROR);
                    dst_len[q] = bytes;
                    if (status) status[q] = st;
                    if (st != AEC_OK) worst = st;
                }
                copy_all(jobs);
                i = j;
            }
            if (batch_sync(k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
            return worst;
        }
    }
    const size_t slot = aec_gpu_encode_bound(&gp, largest);
    if (!k.d_in.ensure(total_in + 32) || !k.d_out.ensure(n * slot) || !k.d_off.ensure(n * sizeof(aec_gpu_enc_result) + 64))
        return AEC_FAIL(AEC_MEM_ERROR);
    // chunk i = [off[i], off[i] + whole samples): the padding between chunks is not input.  Small chunks go up
    // together through pinned staging (one transfer per piece); large ones straight from the callers' buffers, upload
    // and coding alternating so that the kernels of one chunk run while the host stages the next one's copy.
    aec_gpu_enc_result *d_res = static_cast<aec_gpu_enc_result *>(k.d_off.p);
    if (aec_gpu_reserve(k.ctx, &gp, largest) != RC_OK) return AEC_FAIL(AEC_MEM_ERROR);
    const bool stage_in = n >= 16 && largest <= ((size_t)256 << 10) && stage_ensure(k, total_in < kStagePiece ? total_in : kStagePiece);
    size_t staged_to = 0;                                // chunks [.., staged_to) are on the device
    for (size_t i = 0; i < n; i++) {
        const size_t whole = src_len[i] - src_len[i] % c.bytes;
        if (stage_in) {
            if (i == staged_to) {
                size_t j = i;
                while (j < n && off[j + 1] - off[i] <= k.h_stage_cap) j++;
                if (j == i) return AEC_FAIL(AEC_MEM_ERROR);                       // (cannot happen: a chunk fits)
                if (i && batch_sync(k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);   // staging is free again
                std::vector<CopyJob> jobs;
                for (size_t q = i; q < j; q++)
                    jobs.push_back(CopyJob{k.h_stage + (off[q] - off[i]), src[q], src_len[q] - src_len[q] % c.bytes});
                copy_all(jobs);
                if (hipMemcpyAsync(static_cast<uint8_t *>(k.d_in.p) + off[i], k.h_stage, (size_t)(off[j] - off[i]),
                                   hipMemcpyHostToDevice, k.stream) != hipSuccess)
                    return AEC_FAIL(AEC_MEM_ERROR);
                staged_to = j;
            }
        } else if (whole && hipMemcpyAsync(static_cast<uint8_t *>(k.d_in.p) + off[i], src[i], whole, hipMemcpyHostToDevice,
                                           k.stream) != hipSuccess) {
            return AEC_FAIL(AEC_MEM_ERROR);
        }
        const uint64_t pair[2] = {off[i], off[i] + whole};
        rc = aec_gpu_encode_batch_async(k.ctx, &gp, k.d_in.p, pair, 1, static_cast<uint8_t *>(k.d_out.p) + i * slot, slot,
                                        d_res + i, k.stream);
        if (rc != RC_OK) return AEC_FAIL(rc);
    }
    std::vector<aec_gpu_enc_result> res(n);
    if (hipMemcpyAsync(res.data(), d_res, n * sizeof(aec_gpu_enc_result), hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
        batch_sync(k.stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    int worst = AEC_OK;
    // many small streams: whole slots through pinned staging, piece by piece (a transfer per piece beats a copy
    // call per stream even though a slot is the worst case of its stream)
    const size_t per_piece = slot <= kStagePiece ? kStagePiece / slot : 0;
    const bool stage_out = n >= 16 && slot <= ((size_t)512 << 10) && per_piece &&
                           stage_ensure(k, (n < per_piece ? n : per_piece) * slot);
    std::vector<CopyJob> out_jobs;
    for (size_t i = 0; i < n; i++) {
        size_t bytes = (size_t)((res[i].total_bits + 7) / 8);
        if (bytes == 0) bytes = 1;                                           // an empty stream is one zero byte
        int st = AEC_OK;
        if (res[i].overflow) st = AEC_MEM_ERROR;
        else if (bytes > dst_len[i]) { st = AEC_STREAM_ERROR; bytes = dst_len[i]; }   // as aec_buffer_encode: a prefix
        if (stage_out) {
            const size_t first = i - i % per_piece;
            if (i == first) {
                const size_t cnt = n - first < per_piece ? n - first : per_piece;
                if (hipMemcpyAsync(k.h_stage, static_cast<uint8_t *>(k.d_out.p) + first * slot, cnt * slot,
                                   hipMemcpyDeviceToHost, k.stream) != hipSuccess ||
                    batch_sync(k.stream) != hipSuccess)
                    return AEC_FAIL(AEC_MEM_ERROR);
                out_jobs.clear();
            }
            out_jobs.push_back(CopyJob{dst[i], k.h_stage + (i - first) * slot, bytes});
            if (i + 1 == n || (i + 1) % per_piece == 0) copy_all(out_jobs);
        } else if (bytes && hipMemcpyAsync(dst[i], static_cast<uint8_t *>(k.d_out.p) + i * slot, bytes, hipMemcpyDeviceToHost,
                                           k.stream) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        dst_len[i] = bytes;
        if (status) status[i] = st;
        if (st != AEC_OK) worst = st;
    }
    if (batch_sync(k.stream) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
    return worst;
}

}  // namespace

extern "C" {

int aec_buffer_decode_batch(const struct aec_stream *params, size_t n, const void *const *src, const size_t *src_len,
                            void *const *dst, size_t *dst_len, int *status)
{
    try {
        size_t total = 0;
        for (size_t i = 0; i < n; i++) total += dst_len[i];
        return run_parts(n, total, [&](size_t lo, size_t hi) {
            return decode_batch(params, hi - lo, src + lo, src_len + lo, dst + lo, dst_len + lo, status ? status + lo : nullptr);
        });
    } catch (const std::bad_alloc &) { return AEC_MEM_ERROR; }
}

int aec_buffer_encode_batch(const struct aec_stream *params, size_t n, const void *const *src, const size_t *src_len,
                            void *const *dst, size_t *dst_len, int *status)
{
    try {
        size_t total = 0;
        for (size_t i = 0; i < n; i++) total += src_len[i];
        return run_parts(n, total, [&](size_t lo, size_t hi) {
            return encode_batch_host(params, hi - lo, src + lo, src_len + lo, dst + lo, dst_len + lo, status ? status + lo : nullptr);
        });
    } catch (const std::bad_alloc &) { return AEC_MEM_ERROR; }
}

int aec_encode_init(struct aec_stream *strm)
{
    try { return init_common(strm, true); } catch (const std::bad_alloc &) { return AEC_MEM_ERROR; }
}
int aec_decode_init(struct aec_stream *strm)
{
    try { return init_common(strm, false); } catch (const std::bad_alloc &) { return AEC_MEM_ERROR; }
}

int aec_encode(struct aec_stream *strm, int flush)
{
    try {
        return encode_call(strm, flush);
    } catch (const std::bad_alloc &) {
        strm->total_in -= strm->avail_in;      // (added on entry, as on every other way out)
        strm->total_out -= strm->avail_out;
        return AEC_MEM_ERROR;
    }
}

int aec_encode_end(struct aec_stream *strm)
{
    internal_state *s = strm->state;
    int status = AEC_OK;
    if (s->flush == AEC_FLUSH && !s->flushed) status = AEC_STREAM_ERROR;   // reference encode.c:944-945
    free_state(s);
    strm->state = nullptr;
    return status;
}

int aec_buffer_encode(struct aec_stream *strm)
{
    int status = aec_encode_init(strm);
    if (status != AEC_OK) return status;
    status = aec_encode(strm, AEC_FLUSH);
    if (status != AEC_OK) {
        free_state(strm->state);
        strm->state = nullptr;
        return status;
    }
    return aec_encode_end(strm);
}

int aec_decode(struct aec_stream *strm, int flush)
{
    try {
        return decode_call(strm, flush);   // (flush is ignored by the reference, decode.c:797; here it only
                                           // says that the caller is not trickling input in)
    } catch (const std::bad_alloc &) {
        if (strm->state->copy_pending) {       // (nothing of ours may still write the caller's buffer)
            strm->state->copy_pending = false;
            (void)hipStreamSynchronize(strm->state->copy_stream);
        }
        strm->total_in -= strm->avail_in;      // (added on entry, as on every other way out)
        strm->total_out -= strm->avail_out;
        return AEC_MEM_ERROR;
    }
}

int aec_decode_end(struct aec_stream *strm)
{
    free_state(strm->state);
    strm->state = nullptr;
    return AEC_OK;
}

int aec_buffer_decode(struct aec_stream *strm)
{
    int status = aec_decode_init(strm);
    if (status != AEC_OK) return status;
    status = aec_decode(strm, AEC_FLUSH);
    aec_decode_end(strm);
    return status;
}

}  // extern "C"
