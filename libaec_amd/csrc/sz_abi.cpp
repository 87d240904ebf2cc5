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

int compress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    if (param->pixels_per_block <= 0 || param->pixels_per_scanline <= 0) return SZ_PARAM_ERROR;
    struct aec_stream strm;
    const bool planes = param->bits_per_pixel == 32 || param->bits_per_pixel == 64;   // sz_compat.c:134
    strm.bits_per_sample = planes ? 8u : (unsigned)param->bits_per_pixel;
    const Geometry g = geometry(param, strm.bits_per_sample);
    strm.block_size = g.block;
    strm.rsi = g.rsi;
    strm.flags = AEC_NOT_ENFORCE | coder_flags(param->options_mask);                   // sz_compat.c:128
    strm.next_out = static_cast<unsigned char *>(dest);
    strm.avail_out = *destLen;

    // Only whole pixels are coded.  (The reference sizes its padding buffer from the floor of
    // sourceLen / pixel size but copies sourceLen bytes, sz_compat.c:148-166: a trailing fraction of
    // a pixel overruns that buffer there.  It is dropped here.)
    sourceLen -= sourceLen % (planes ? (size_t)param->bits_per_pixel / 8 : g.pixel);
    std::vector<uint8_t> plane_buf;
    const uint8_t *src = static_cast<const uint8_t *>(source);
    if (planes) {
        plane_buf.resize(sourceLen);
        to_planes(plane_buf.data(), src, sourceLen, (size_t)param->bits_per_pixel / 8);
        src = plane_buf.data();
    }

    // every scan line becomes one RSI: pad it to whole blocks, repeating the last pixel when the
    // preprocessor is on and with zero pixels otherwise (sz_compat.c:71-94, 148-166)
    const size_t lines = (sourceLen / g.pixel + (size_t)param->pixels_per_scanline - 1) /
                         (size_t)param->pixels_per_scanline;
    std::vector<uint8_t> padded(g.padded_line * lines);
    const bool repeat = (strm.flags & AEC_DATA_PREPROCESS) != 0;
    size_t in = 0, out = 0;
    while (in < sourceLen) {
        const size_t take = sourceLen - in < g.line ? sourceLen - in : g.line;
        memcpy(padded.data() + out, src + in, take);
        in += take;
        const uint8_t *fill = src + in - g.pixel;
        for (size_t k = take; k < g.padded_line; k += g.pixel) {
            if (repeat) memcpy(padded.data() + out + k, fill, g.pixel);
            else memset(padded.data() + out + k, 0, g.pixel);
        }
        out += g.padded_line;
    }
    strm.next_in = padded.data();
    strm.avail_in = padded.size();

    const int rc = aec_buffer_encode(&strm);
    *destLen = strm.total_out;                                                          // sz_compat.c:175
    return rc == AEC_STREAM_ERROR ? SZ_OUTBUFF_FULL : rc;                               // sz_compat.c:171-174
}

int decompress(void *dest, size_t *destLen, const void *source, size_t sourceLen, SZ_com_t *param)
{
    if (param->pixels_per_block <= 0 || param->pixels_per_scanline <= 0) return SZ_PARAM_ERROR;
    struct aec_stream strm;
    const bool planes = param->bits_per_pixel == 32 || param->bits_per_pixel == 64;
    strm.bits_per_sample = planes ? 8u : (unsigned)param->bits_per_pixel;
    const Geometry g = geometry(param, strm.bits_per_sample);
    strm.block_size = g.block;
    strm.rsi = g.rsi;
    strm.flags = coder_flags(param->options_mask);                                      // sz_compat.c:205
    strm.next_in = static_cast<const unsigned char *>(source);
    strm.avail_in = sourceLen;

    const bool padded_lines = param->pixels_per_scanline % param->pixels_per_block != 0;
    size_t lines = 0;
    std::vector<uint8_t> tmp;
    if (padded_lines || planes) {                                                       // sz_compat.c:222-236
        if (padded_lines) {
            lines = (*destLen / g.pixel + (size_t)param->pixels_per_scanline - 1) /
                    (size_t)param->pixels_per_scanline;
            tmp.resize(g.padded_line * lines);
        } else {
            tmp.resize(*destLen);
        }
        strm.next_out = tmp.data();
        strm.avail_out = tmp.size();
    } else {
        strm.next_out = static_cast<unsigned char *>(dest);
        strm.avail_out = *destLen;
    }

    const int rc = aec_buffer_decode(&strm);
    if (rc != AEC_OK) return rc;

    size_t total = strm.total_out;
    if (padded_lines) {                                                                 // sz_compat.c:96-108
        size_t w = g.line;
        for (size_t r = g.padded_line; r < strm.total_out; r += g.padded_line) {
            memmove(tmp.data() + w, tmp.data() + r, g.line);
            w += g.line;
        }
        total = lines * g.line;
    }
    if (total < *destLen) *destLen = total;                                             // sz_compat.c:256-257
    if (planes) from_planes(static_cast<uint8_t *>(dest), tmp.data(), *destLen, (size_t)param->bits_per_pixel / 8);
    else if (padded_lines) memcpy(dest, tmp.data(), *destLen);
    return SZ_OK;
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

int SZ_encoder_enabled(void) { return 1; }

char SZ_Compress(void) { return SZ_OK; }

}  // extern "C"
