// aec_abi.cpp -- the libaec C ABI (include/libaec.h) on top of the device-resident batch API.
//
// This is the host-side stream layer the reference implements as two resumable finite state
// machines (reference src/encode.c:467-518, 661-754, 909-963 and src/decode.c:342-400,
// 797-854).  Its contract is kept -- any chunking of input and output, cumulative
// total_in/total_out, AEC_FLUSH semantics, the same return codes -- but the states are replaced
// by staging: input is collected until whole RSIs are available, every whole RSI present at a
// call is coded in ONE batch on the GPU with the bit position and k carried between batches,
// and produced bytes wait in a queue until the caller offers room.  There is no CPU codec in
// here: without a working HIP device every call fails with AEC_MEM_ERROR.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <vector>

#include "../../include/aec_gpu.h"
#include "../../include/libaec.h"
#include "aec_cfg.h"

using namespace aec;

namespace {

// AEC_ABI_TRACE=1 in the environment: say on stderr where a device-side failure was detected
int fail_at(int code, int line)
{
    static const bool trace = getenv("AEC_ABI_TRACE") != nullptr;
    if (trace) fprintf(stderr, "libaec (MI355X): error %d raised at aec_abi.cpp:%d (last HIP error: %s)\n", code, line,
                       hipGetErrorString(hipPeekAtLastError()));
    return code;
}
#define AEC_FAIL(code) fail_at((code), __LINE__)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n)
    {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        want = (want + 255) & ~(size_t)255;
        if (hipMalloc(&p, want) != hipSuccess) return false;
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

}  // namespace

struct internal_state {
    bool encoder;
    aec_gpu_params prm;
    Cfg cfg;                       // derived values (sizes are per batch, not used from here)
    int device;
    aec_gpu_ctx *ctx;
    hipStream_t stream;
    DevBuf d_in, d_out, d_off, d_res;
    void *h_res;                   // pinned mirror of the device result record

    size_t out_room;               // decoder: avail_out of the current call (sizes the index look-ahead)
    uint64_t rsi_bits_seen;        // decoder: average coded RSI of the previous batch (0 = none yet)

    std::vector<uint8_t> stage;    // input not yet coded
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

    // decoder position: the stage holds the stream from byte `stage_base`; the RSI being
    // decoded starts at absolute bit rsi_start_bit; `delivered` samples of it were queued
    uint64_t stage_base;
    uint64_t rsi_start_bit;
    uint64_t delivered;
    bool new_input;
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
    DevBuf d_in, d_out, d_off, d_res;
    void *h_res = nullptr;
};
constexpr size_t kPoolMax = 8, kKeepBytes = (size_t)64 << 20;
std::mutex g_pool_mu;
std::vector<Kit> *g_pool = nullptr;      // heap object on purpose: no destructor at exit

void destroy_kit(Kit &k)
{
    k.d_in.release();
    k.d_out.release();
    k.d_off.release();
    k.d_res.release();
    if (k.h_res) (void)hipHostFree(k.h_res);
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

void park_kit(Kit &k)
{
    for (DevBuf *b : {&k.d_in, &k.d_out, &k.d_off})
        if (b->cap > kKeepBytes) b->release();
    aec_gpu_trim(k.ctx, kKeepBytes);
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (!g_pool) g_pool = new (std::nothrow) std::vector<Kit>();
        if (g_pool && g_pool->size() < kPoolMax) {
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
    k.h_res = s->h_res;
    // everything enqueued for this stream object has been waited for by the calls that enqueued it
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
    s->outq_pos = 0;
    s->k = 0;
    s->part_bits = 0;
    s->part_byte = 0;
    s->any_bits = false;
    s->finished = false;
    s->flush = AEC_NO_FLUSH;
    s->flushed = false;
    s->stage_base = 0;
    s->rsi_start_bit = 0;
    s->delivered = 0;
    s->new_input = false;
    s->sticky_error = AEC_OK;
    s->ctx = nullptr;
    s->stream = nullptr;
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
        s->h_res = k.h_res;
        aec_gpu_set_index_hint(s->ctx, 0);
    } else if (s->device < 0 || aec_gpu_create(&s->ctx) != RC_OK || hipStreamCreate(&s->stream) != hipSuccess ||
               hipHostMalloc(&s->h_res, 256, hipHostMallocDefault) != hipSuccess || !s->d_res.ensure(256)) {
        free_state(s);
        return AEC_FAIL(AEC_MEM_ERROR);   // no usable HIP device: the product has no CPU path
    }
    strm->state = s;
    strm->total_in = 0;         // reference encode.c:897-898, decode.c:785-786
    strm->total_out = 0;
    return AEC_OK;
}

size_t drain(struct aec_stream *strm, internal_state *s, size_t granule)
{
    size_t n = s->outq.size() - s->outq_pos;
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

// Code `nbytes` of staged input (whole samples) as one GPU batch; append produced whole bytes
// to the queue and keep the open byte / k as carry.
// With `strm` given and nothing queued, finished bytes go straight into the caller's buffer (no
// pass through the queue); only what does not fit, and the open byte, are queued.
int encode_batch(internal_state *s, const uint8_t *data, size_t nbytes, struct aec_stream *strm = nullptr)
{
    if (nbytes == 0) return AEC_OK;
    const size_t cap = aec_gpu_encode_bound(&s->prm, nbytes);
    const uint64_t nrsi = aec_gpu_rsi_count(&s->prm, nbytes);
    (void)nrsi;
    if (!s->d_in.ensure(nbytes + 16) || !s->d_out.ensure(cap)) return AEC_FAIL(AEC_MEM_ERROR);
    if (hipMemcpyAsync(s->d_in.p, data, nbytes, hipMemcpyHostToDevice, s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const int rc = aec_gpu_encode_async(s->ctx, &s->prm, s->d_in.p, nbytes, s->d_out.p, cap, s->part_bits,
                                        s->k, nullptr, static_cast<aec_gpu_enc_result *>(s->d_res.p),
                                        s->stream);
    if (rc != RC_OK) return AEC_FAIL(rc);
    if (hipMemcpyAsync(s->h_res, s->d_res.p, sizeof(aec_gpu_enc_result), hipMemcpyDeviceToHost,
                       s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const aec_gpu_enc_result res = *static_cast<aec_gpu_enc_result *>(s->h_res);
    if (res.overflow) return AEC_FAIL(AEC_MEM_ERROR);   // cannot happen: cap is the worst case
    const uint64_t bits = (uint64_t)s->part_bits + res.total_bits;
    const size_t whole = (size_t)(bits / 8), nb = (size_t)((bits + 7) / 8);
    size_t direct = 0;
    if (strm && s->outq.empty()) direct = whole < strm->avail_out ? whole : strm->avail_out;
    const uint8_t *d_bytes = static_cast<const uint8_t *>(s->d_out.p);
    if (direct && hipMemcpy(strm->next_out, d_bytes, direct, hipMemcpyDeviceToHost) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const size_t at = s->outq.size();
    s->outq.resize(at + (nb - direct));
    if (nb > direct &&
        hipMemcpy(s->outq.data() + at, d_bytes + direct, nb - direct, hipMemcpyDeviceToHost) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    if (direct) strm->next_out[0] |= s->part_byte;
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

// Decode everything decodable in the staged input (see internal_state for the cursor).
// Samples go straight into the caller's buffer as far as it has room (the queue is empty whenever
// this runs); the rest is queued.
int decode_staged(internal_state *s, struct aec_stream *strm)
{
    const Cfg &c = s->cfg;
    const size_t nbytes = s->stage.size();
    if (nbytes == 0) return AEC_OK;
    const uint64_t start_rel = s->rsi_start_bit - s->stage_base * 8;
    const uint64_t avail_bits = (uint64_t)nbytes * 8 - start_rel;
    // the shortest possible RSI is all zero blocks: one zero-run CDS (id_len + 2 bits) per segment,
    // plus the reference sample when the preprocessor is on; that bounds the offset table
    const uint64_t min_rsi_bits = (uint64_t)c.segs_per_rsi * (c.id_len + 2) + ((c.flags & F_PREPROCESS) ? c.bps : 0);
    const uint64_t max_rsi = avail_bits / min_rsi_bits + 2;
    if (!s->d_in.ensure(nbytes + 16) || !s->d_off.ensure((max_rsi + 1) * 8)) return AEC_FAIL(AEC_MEM_ERROR);
    if (hipMemcpyAsync(s->d_in.p, s->stage.data(), nbytes, hipMemcpyHostToDevice, s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    aec_gpu_dec_result *dres = static_cast<aec_gpu_dec_result *>(s->d_res.p);
    aec_gpu_dec_result *hres = static_cast<aec_gpu_dec_result *>(s->h_res);
    // Look-ahead of the speculative index = a small multiple of the average coded RSI: measured on
    // the previous batch of this stream, else estimated from the room the caller offers for output.
    uint64_t hint = s->rsi_bits_seen;
    if (!hint) {
        const uint64_t rsi_bytes = (uint64_t)c.rsi * c.bs * c.bytes;
        const uint64_t expect = (s->out_room + rsi_bytes - 1) / rsi_bytes;
        if (expect) hint = avail_bits / expect;
    }
    aec_gpu_set_index_hint(s->ctx, hint + hint / 2);
    int rc = aec_gpu_index_async(s->ctx, &s->prm, s->d_in.p, nbytes, start_rel,
                                 static_cast<uint64_t *>(s->d_off.p), max_rsi, dres, s->stream);
    if (rc != RC_OK) return AEC_FAIL(rc);
    if (hipMemcpyAsync(hres, dres, sizeof(*hres), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
        hipStreamSynchronize(s->stream) != hipSuccess)
        return AEC_FAIL(AEC_MEM_ERROR);
    const aec_gpu_dec_result idx = *hres;
    if (idx.n_rsi) s->rsi_bits_seen = (idx.end_bit - start_rel) / idx.n_rsi;
    const uint64_t n_items = idx.n_rsi + (idx.tail_blocks ? 1 : 0);
    const uint64_t blocks = idx.n_rsi * c.rsi + idx.tail_blocks;
    const size_t blk_bytes = (size_t)c.bs * c.bytes;
    if (blocks * c.bs > s->delivered) {
        if (!s->d_out.ensure(blocks * blk_bytes + 16)) return AEC_FAIL(AEC_MEM_ERROR);
        rc = aec_gpu_decode_async(s->ctx, &s->prm, s->d_in.p, nbytes, static_cast<uint64_t *>(s->d_off.p),
                                  n_items, blocks, s->d_out.p, dres, s->stream);
        if (rc != RC_OK) return AEC_FAIL(rc);
        if (hipMemcpyAsync(hres, dres, sizeof(*hres), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
            hipStreamSynchronize(s->stream) != hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        if (hres->status == DEC_DATA_ERROR) return AEC_DATA_ERROR;
        if (hres->status != DEC_OK) return AEC_DATA_ERROR;   // the index pass vouched for completeness
        const size_t skip = (size_t)s->delivered * c.bytes;
        const size_t total = (size_t)blocks * blk_bytes;
        const uint8_t *d_bytes = static_cast<const uint8_t *>(s->d_out.p) + skip;
        size_t direct = 0;
        if (s->outq.empty()) {
            direct = total - skip < strm->avail_out ? total - skip : strm->avail_out;
            direct -= direct % c.bytes;
        }
        if (direct) {
            if (hipMemcpy(strm->next_out, d_bytes, direct, hipMemcpyDeviceToHost) != hipSuccess) return AEC_FAIL(AEC_MEM_ERROR);
            strm->next_out += direct;
            strm->avail_out -= direct;
        }
        if (total - skip > direct) {
            const size_t at = s->outq.size();
            s->outq.resize(at + (total - skip - direct));
            if (hipMemcpy(s->outq.data() + at, d_bytes + direct, total - skip - direct, hipMemcpyDeviceToHost) !=
                hipSuccess)
                return AEC_FAIL(AEC_MEM_ERROR);
        }
    }
    // advance the cursor to the start of the (possibly empty) trailing partial RSI
    uint64_t new_start_rel = idx.end_bit;
    if (idx.tail_blocks) {
        uint64_t off = 0;
        if (hipMemcpy(&off, static_cast<uint64_t *>(s->d_off.p) + idx.n_rsi, 8, hipMemcpyDeviceToHost) !=
            hipSuccess)
            return AEC_FAIL(AEC_MEM_ERROR);
        new_start_rel = off;
        s->delivered = idx.tail_blocks * c.bs;
    } else {
        s->delivered = 0;
    }
    s->rsi_start_bit = s->stage_base * 8 + new_start_rel;
    const uint64_t drop = s->rsi_start_bit / 8 - s->stage_base;
    if (drop) {
        s->stage.erase(s->stage.begin(), s->stage.begin() + (ptrdiff_t)drop);
        s->stage_base += drop;
    }
    return idx.status == DEC_DATA_ERROR ? AEC_DATA_ERROR : AEC_OK;
}

}  // namespace

extern "C" {

int aec_encode_init(struct aec_stream *strm) { return init_common(strm, true); }
int aec_decode_init(struct aec_stream *strm) { return init_common(strm, false); }

int aec_encode(struct aec_stream *strm, int flush)
{
    internal_state *s = strm->state;
    const size_t bytes = s->cfg.bytes;
    const size_t rsi_bytes = (size_t)s->cfg.rsi * s->cfg.bs * bytes;
    s->flush = flush;
    strm->total_in += strm->avail_in;     // reference encode.c:919-920
    strm->total_out += strm->avail_out;

    for (;;) {
        drain(strm, s, 1);
        if (!s->outq.empty()) break;      // output full
        if (s->finished) {
            s->flushed = true;            // reference encode.c:689-694
            break;
        }
        // only whole samples are ever consumed (reference encode.c:673-674)
        size_t take = strm->avail_in - strm->avail_in % bytes;
        int rc = AEC_OK;
        if (s->stage.empty() && take >= rsi_bytes) {
            // whole RSIs offered and nothing staged: code them from the caller's buffer
            const size_t direct = take / rsi_bytes * rsi_bytes;
            rc = encode_batch(s, strm->next_in, direct, strm);
            strm->next_in += direct;
            strm->avail_in -= direct;
            if (rc != AEC_OK) {
                strm->total_in -= strm->avail_in;
                strm->total_out -= strm->avail_out;
                return rc;
            }
            continue;
        }
        if (take) {
            s->stage.insert(s->stage.end(), strm->next_in, strm->next_in + take);
            strm->next_in += take;
            strm->avail_in -= take;
        }
        const size_t whole = s->stage.size() / rsi_bytes * rsi_bytes;
        if (whole) {
            rc = encode_batch(s, s->stage.data(), whole, strm);
            s->stage.erase(s->stage.begin(), s->stage.begin() + (ptrdiff_t)whole);
        } else if (flush == AEC_FLUSH) {
            // last, short RSI (reference encode.c:676-684), then the final byte (686-695)
            rc = encode_batch(s, s->stage.data(), s->stage.size(), strm);
            s->stage.clear();
            if (rc == AEC_OK) {
                if (s->part_bits || !s->any_bits) s->outq.push_back(s->part_byte);
                s->part_bits = 0;
                s->part_byte = 0;
                s->finished = true;
            }
        } else {
            break;                        // need more input
        }
        if (rc != AEC_OK) {
            strm->total_in -= strm->avail_in;
            strm->total_out -= strm->avail_out;
            return rc;
        }
    }
    strm->total_in -= strm->avail_in;     // reference encode.c:933-934
    strm->total_out -= strm->avail_out;
    return AEC_OK;
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
    (void)flush;                          // ignored by the reference as well (decode.c:797)
    internal_state *s = strm->state;
    const size_t bytes = s->cfg.bytes;
    strm->total_in += strm->avail_in;     // reference decode.c:811-812
    strm->total_out += strm->avail_out;
    int rc = s->sticky_error;

    while (rc == AEC_OK) {
        drain(strm, s, bytes);
        if (!s->outq.empty()) break;      // output full (or less than one sample of room)
        if (strm->avail_in) {
            s->stage.insert(s->stage.end(), strm->next_in, strm->next_in + strm->avail_in);
            strm->next_in += strm->avail_in;
            strm->avail_in = 0;
            s->new_input = true;
        }
        if (!s->new_input) break;         // nothing new to look at
        s->new_input = false;
        s->out_room = strm->avail_out;
        rc = decode_staged(s, strm);
        if (rc == AEC_DATA_ERROR) s->sticky_error = rc;
    }
    if (rc != AEC_OK) return rc;          // reference decode.c:818-819 (totals left as they are)
    if (strm->avail_out > 0 && strm->avail_out < bytes) return AEC_FAIL(AEC_MEM_ERROR);   // decode.c:821-823
    strm->total_in -= strm->avail_in;     // reference decode.c:827-828
    strm->total_out -= strm->avail_out;
    return AEC_OK;
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
