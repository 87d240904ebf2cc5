// sz_abi.cpp -- SZIP entry points (include/szlib.h) over the libaec ABI of this library.
//
// Restates the data marshalling of the reference's shim (reference src/sz_compat.c): option
// mapping (:12-27), byte-plane interleave of 32/64-bit pixels (:39-69), padding of every scan
// line to a whole number of blocks so that one scan line = one RSI (:71-108), then ONE
// aec_buffer_encode / aec_buffer_decode call, which is where the GPU does the work.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/szlib.h"

namespace {

unsigned coder_flags(int sz_options)                      // sz_compat.c:12-27
{
    unsigned f = 0;
    if (sz_options & SZ_MSB_OPTION_MASK) f |= AEC_DATA_MSB;
    if (sz_options & SZ_NN_OPTION_MASK) f |= AEC_DATA_PREPROCESS;
    return f;
}

size_t container_bytes(int bits)                          // sz_compat.c:29-37
{
    return bits > 16 ? 4 : (bits > 8 ? 2 : 1);
}

// pixel-major -> plane-major: byte j of pixel i goes to plane j (sz_compat.c:39-53)
void to_planes(uint8_t *dst, const uint8_t *src, size_t n, size_t word)
{
    const size_t pixels = n / word;
    for (size_t j = 0; j < word; j++) {
        uint8_t *plane = dst + j * pixels;
        for (size_t i = 0; i < pixels; i++) plane[i] = src[i * word + j];
    }
}

// plane-major -> pixel-major (sz_compat.c:55-69)
void from_planes(uint8_t *dst, const uint8_t *src, size_t n, size_t word)
{
    const size_t pixels = n / word;
    for (size_t j = 0; j < word; j++) {
        const uint8_t *plane = src + j * pixels;
        for (size_t i = 0; i < pixels; i++) dst[i * word + j] = plane[i];
    }
}

struct Geometry {
    unsigned block, rsi;
    size_t pixel, line, padded_line;   // bytes
};

Geometry geometry(const SZ_com_t *p, unsigned bits_per_sample)
{
    Geometry g;
    g.block = (unsigned)p->pixels_per_block;
    g.rsi = (unsigned)((p->pixels_per_scanline + p->pixels_per_block - 1) / p->pixels_per_block);
    g.pixel = container_bytes((int)bits_per_sample);
    g.line = (size_t)p->pixels_per_scanline * g.pixel;
    g.padded_line = (size_t)g.rsi * g.block * g.pixel;
    return g;
}

// ---- one chunk, in two steps, so that the batch entry points can put ONE coder call between them ----
// compress: marshal `source` (byte planes, scan-line padding) into `buf`, fill the stream parameters
struct Job {
    struct aec_stream strm;
    Geometry g;
    bool planes, padded_lines;
    size_t lines;
    std::vector<uint8_t> buf;      // compress: the coder's input; decompress: the coder's output when it needs fixing up
    const uint8_t *in = nullptr;   // compress: where the coder's input lies (buf, or the caller's buffer as it is)
    size_t in_len = 0;
};

int prepare_compress(Job &j, const void *source, size_t sourceLen, const SZ_com_t *param)
{
    if (param->pixels_per_block <= 0 || param->pixels_per_scanline <= 0) return SZ_PARAM_ERROR;
    j.planes = param->bits_per_pixel == 32 || param->bits_per_pixel == 64;   // sz_compat.c:134
    j.strm.bits_per_sample = j.planes ? 8u : (unsigned)param->bits_per_pixel;
    j.g = geometry(param, j.strm.bits_per_sample);
    const Geometry &g = j.g;
    j.strm.block_size = g.block;
    j.strm.rsi = g.rsi;
    j.strm.flags = AEC_NOT_ENFORCE | coder_flags(param->options_mask);                   // sz_compat.c:128
    // Only whole pixels are coded.  (The reference sizes its padding buffer from the floor of
    // sourceLen / pixel size but copies sourceLen bytes, sz_compat.c:148-166: a trailing fraction of
    // a pixel overruns that buffer there.  It is dropped here.)
    sourceLen -= sourceLen % (j.planes ? (size_t)param->bits_per_pixel / 8 : g.pixel);
    std::vector<uint8_t> plane_buf;
    const uint8_t *src = static_cast<const uint8_t *>(source);
    if (!j.planes && g.padded_line == g.line && sourceLen % g.line == 0) {
        // whole scan lines of whole blocks (the usual HDF5 chunk): nothing to marshal, the coder reads the
        // caller's buffer
        j.in = src;
        j.in_len = sourceLen;
        return SZ_OK;
    }
    if (j.planes) {
        plane_buf.resize(sourceLen);
        to_planes(plane_buf.data(), src, sourceLen, (size_t)param->bits_per_pixel / 8);
        src = plane_buf.data();
    }
    // every scan line becomes one RSI: pad it to whole blocks, repeating the last pixel when the
    // preprocessor is on and with zero pixels otherwise (sz_compat.c:71-94, 148-166)
    const size_t lines = (sourceLen / g.pixel + (size_t)param->pixels_per_scanline - 1) /
                         (size_t)param->pixels_per_scanline;
    j.buf.assign(g.padded_line * lines, 0);
    const bool repeat = (j.strm.flags & AEC_DATA_PREPROCESS) != 0;
    size_t in = 0, out = 0;
    while (in < sourceLen) {
        const size_t take = sourceLen - in < g.line ? sourceLen - in : g.line;
        memcpy(j.buf.data() + out, src + in, take);
        in += take;
        const uint8_t *fill = src + in - g.pixel;
        for (size_t k = take; k < g.padded_line; k += g.pixel) {
            if (repeat) memcpy(j.buf.data() + out + k, fill, g.pixel);
            else memset(j.buf.data() + out + k, 0, g.pixel);
        }
        out += g.padded_line;
    }
    j.in = j.buf.data();
    j.in_len = j.buf.size();
    return SZ_OK;
}

int compress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    Job j;
    const int prc = prepare_compress(j, source, sourceLen, param);
    if (prc != SZ_OK) return prc;
    j.strm.next_out = static_cast<unsigned char *>(dest);
    j.strm.avail_out = *destLen;
    j.strm.next_in = j.in;
    j.strm.avail_in = j.in_len;
    const int rc = aec_buffer_encode(&j.strm);
    *destLen = j.strm.total_out;                                                        // sz_compat.c:175
    return rc == AEC_STREAM_ERROR ? SZ_OUTBUFF_FULL : rc;                               // sz_compat.c:171-174
}

// decompress: where the coder must write (`out` / `cap`), then the fix-up of what it wrote
int prepare_decompress(Job &j, void *dest, size_t destLen, const SZ_com_t *param, unsigned char *&out, size_t &cap)
{
    if (param->pixels_per_block <= 0 || param->pixels_per_scanline <= 0) return SZ_PARAM_ERROR;
    j.planes = param->bits_per_pixel == 32 || param->bits_per_pixel == 64;
    j.strm.bits_per_sample = j.planes ? 8u : (unsigned)param->bits_per_pixel;
    j.g = geometry(param, j.strm.bits_per_sample);
    j.strm.block_size = j.g.block;
    j.strm.rsi = j.g.rsi;
    j.strm.flags = coder_flags(param->options_mask);                                      // sz_compat.c:205
    j.padded_lines = param->pixels_per_scanline % param->pixels_per_block != 0;
    j.lines = 0;
    if (j.padded_lines || j.planes) {                                                     // sz_compat.c:222-236
        if (j.padded_lines) {
            j.lines = (destLen / j.g.pixel + (size_t)param->pixels_per_scanline - 1) /
                      (size_t)param->pixels_per_scanline;
            j.buf.resize(j.g.padded_line * j.lines);
        } else {
            j.buf.resize(destLen);
        }
        out = j.buf.data();
        cap = j.buf.size();
    } else {
        out = static_cast<unsigned char *>(dest);
        cap = destLen;
    }
    return SZ_OK;
}

void finish_decompress(Job &j, void *dest, size_t *destLen, size_t total_out, const SZ_com_t *param)
{
    const Geometry &g = j.g;
    size_t total = total_out;
    if (j.padded_lines) {                                                                 // sz_compat.c:96-108
        size_t w = g.line;
        for (size_t r = g.padded_line; r < total_out; r += g.padded_line) {
            memmove(j.buf.data() + w, j.buf.data() + r, g.line);
            w += g.line;
        }
        total = j.lines * g.line;
    }
    if (total < *destLen) *destLen = total;                                               // sz_compat.c:256-257
    if (j.planes) from_planes(static_cast<uint8_t *>(dest), j.buf.data(), *destLen, (size_t)param->bits_per_pixel / 8);
    else if (j.padded_lines) memcpy(dest, j.buf.data(), *destLen);
}

int decompress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    Job j;
    unsigned char *out = nullptr;
    size_t cap = 0;
    const int prc = prepare_decompress(j, dest, *destLen, param, out, cap);
    if (prc != SZ_OK) return prc;
    j.strm.next_in = static_cast<const unsigned char *>(source);
    j.strm.avail_in = sourceLen;
    j.strm.next_out = out;
    j.strm.avail_out = cap;
    const int rc = aec_buffer_decode(&j.strm);
    if (rc != AEC_OK) return rc;
    finish_decompress(j, dest, destLen, j.strm.total_out, param);
    return SZ_OK;
}

// ---- n chunks with the same parameters: one coder call for all of them ----------------------------------
int batch_compress(void *const *dest, size_t *destLen, const void *const *source, const size_t *sourceLen, size_t n,
                   SZ_com_t *param, int *status)
{
    std::vector<Job> jobs(n);
    std::vector<const void *> src(n);
    std::vector<size_t> src_len(n);
    for (size_t i = 0; i < n; i++) {
        const int prc = prepare_compress(jobs[i], source[i], sourceLen[i], param);
        if (prc != SZ_OK) return prc;
        src[i] = jobs[i].in;
        src_len[i] = jobs[i].in_len;
    }
    std::vector<int> st(n, AEC_OK);
    const int rc = aec_buffer_encode_batch(&jobs[0].strm, n, src.data(), src_len.data(), dest, destLen, st.data());
    int worst = SZ_OK;
    for (size_t i = 0; i < n; i++) {
        const int one = st[i] == AEC_STREAM_ERROR ? SZ_OUTBUFF_FULL : st[i];
        if (status) status[i] = one;
        if (one != SZ_OK) worst = one;
    }
    return (rc != AEC_OK && worst == SZ_OK) ? rc : worst;
}

int batch_decompress(void *const *dest, size_t *destLen, const void *const *source, const size_t *sourceLen, size_t n,
                     SZ_com_t *param, int *status)
{
    std::vector<Job> jobs(n);
    std::vector<void *> out(n);
    std::vector<size_t> cap(n);
    for (size_t i = 0; i < n; i++) {
        unsigned char *o = nullptr;
        const int prc = prepare_decompress(jobs[i], dest[i], destLen[i], param, o, cap[i]);
        if (prc != SZ_OK) return prc;
        out[i] = o;
    }
    std::vector<int> st(n, AEC_OK);
    const int rc = aec_buffer_decode_batch(&jobs[0].strm, n, source, sourceLen, out.data(), cap.data(), st.data());
    int worst = SZ_OK;
    for (size_t i = 0; i < n; i++) {
        if (st[i] == AEC_OK) finish_decompress(jobs[i], dest[i], &destLen[i], cap[i], param);
        if (status) status[i] = st[i];
        if (st[i] != AEC_OK) worst = st[i];
    }
    return (rc != AEC_OK && worst == SZ_OK) ? rc : worst;
}

}  // namespace

extern "C" {

// (allocation failures of the marshalling buffers must not leave through the C entry points)
int SZ_BufftoBuffCompress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    try { return compress(dest, destLen, source, sourceLen, param); } catch (const std::bad_alloc &) { return SZ_MEM_ERROR; }
}

int SZ_BufftoBuffDecompress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    try { return decompress(dest, destLen, source, sourceLen, param); } catch (const std::bad_alloc &) { return SZ_MEM_ERROR; }
}

int SZ_BatchCompress(void *const *dest, size_t *destLen, const void *const *source, const size_t *sourceLen, size_t n,
                     SZ_com_t *param, int *status)
{
    if (n == 0) return SZ_OK;
    try { return batch_compress(dest, destLen, source, sourceLen, n, param, status); } catch (const std::bad_alloc &) { return SZ_MEM_ERROR; }
}

int SZ_BatchDecompress(void *const *dest, size_t *destLen, const void *const *source, const size_t *sourceLen, size_t n,
                       SZ_com_t *param, int *status)
{
    if (n == 0) return SZ_OK;
    try { return batch_decompress(dest, destLen, source, sourceLen, n, param, status); } catch (const std::bad_alloc &) { return SZ_MEM_ERROR; }
}

int SZ_encoder_enabled(void) { return 1; }

char SZ_Compress(void) { return SZ_OK; }

}  // extern "C"
