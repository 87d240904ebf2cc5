/*
 * libaec.h -- C ABI of the MI355X-native adaptive entropy coder.
 *
 * Binary compatible with erget/libaec 0.3.4: the stream structure (reference
 * src/libaec.h:67-97), the flag, return-code and flush constants (:105-149) and the eight
 * entry points (:154-166) have the same names, layout, argument meaning and error behaviour,
 * so a program built against the reference header runs against this library unchanged
 * (shared object name libaec.so.0).  The work behind the entry points is done by HIP kernels
 * on the current HIP device; see DESIGN.md and INTEGRATION.md.
 */
#ifndef LIBAEC_H
#define LIBAEC_H 1

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

struct internal_state;

/* reference src/libaec.h:67-97 -- field order and types are the ABI */
struct aec_stream {
    const unsigned char *next_in;   /* next input byte */
    size_t avail_in;                /* bytes available at next_in */
    size_t total_in;                /* bytes consumed so far */
    unsigned char *next_out;        /* where the next output byte goes */
    size_t avail_out;               /* free bytes at next_out */
    size_t total_out;               /* bytes produced so far */
    unsigned int bits_per_sample;   /* 1..32 */
    unsigned int block_size;        /* samples per block: 8, 16, 32, 64 (any even value <= 64
                                       with AEC_NOT_ENFORCE).  LIMIT of this library: the reference accepts
                                       any even size with AEC_NOT_ENFORCE (src/encode.c:780-783) although
                                       sizes above 64 overrun its own coded-data-set buffer (encode.h:64-66);
                                       here they are AEC_CONF_ERROR -- one lane owns one block */
    unsigned int rsi;               /* blocks per reference sample interval, 1..4096 */
    unsigned int flags;             /* AEC_DATA_* | AEC_RESTRICTED | AEC_PAD_RSI | AEC_NOT_ENFORCE */
    struct internal_state *state;   /* owned by the library between *_init and *_end */
};

/* sample description flags, reference src/libaec.h:105-124 */
#define AEC_DATA_SIGNED 1      /* samples are two's complement */
#define AEC_DATA_3BYTE 2       /* 17..24 bit samples are stored in 3 bytes instead of 4 */
#define AEC_DATA_MSB 4         /* most significant byte first (default: least significant first) */
#define AEC_DATA_PREPROCESS 8  /* run the unit-delay predictor / sign mapper */
#define AEC_RESTRICTED 16      /* restricted code-option set (bits_per_sample <= 4 only) */
#define AEC_PAD_RSI 32         /* decoder: every RSI starts on a byte boundary */
#define AEC_NOT_ENFORCE 64     /* allow non-standard (even) block sizes */

/* return codes, reference src/libaec.h:129-133 */
#define AEC_OK 0
#define AEC_CONF_ERROR (-1)
#define AEC_STREAM_ERROR (-2)
#define AEC_DATA_ERROR (-3)
#define AEC_MEM_ERROR (-4)

/* flush modes, reference src/libaec.h:141-149 */
#define AEC_NO_FLUSH 0   /* more input may follow */
#define AEC_FLUSH 1      /* this is the last input: drain everything and pad the final byte */

#define LIBAEC_API __attribute__((visibility("default")))

/* streaming interface, reference src/libaec.h:154-160 (encode.c:773-948, decode.c:694-841)
 *
 * WHAT the calls deliver is the reference's, byte for byte; WHEN may differ, because work goes to the device
 * in batches:
 *   aec_encode(AEC_NO_FLUSH) codes the whole RSIs it holds once 1 MiB is staged, or on a call that brings no new
 *     input, or on AEC_FLUSH -- until then avail_out may stay untouched although input was taken.
 *   aec_decode runs a batch on every call EXCEPT for a caller that feeds at most 8 new bytes per call with
 *     AEC_NO_FLUSH while fewer than 4 KiB wait: those bytes are decoded by the next call that brings nothing new
 *     (avail_in == 0) or passes AEC_FLUSH.  A caller that stops as soon as its input is used up must make that
 *     last call (the reference's own callers do: src/aec.c:191-221, tests/check_aec.c:138-166), or pass
 *     AEC_FLUSH (which the reference ignores for decoding, decode.c:797) with its last piece of input.
 * The library has no CPU codec: without a HIP device every call returns AEC_MEM_ERROR. */
LIBAEC_API int aec_encode_init(struct aec_stream *strm);
LIBAEC_API int aec_encode(struct aec_stream *strm, int flush);
LIBAEC_API int aec_encode_end(struct aec_stream *strm);
LIBAEC_API int aec_decode_init(struct aec_stream *strm);
LIBAEC_API int aec_decode(struct aec_stream *strm, int flush);
LIBAEC_API int aec_decode_end(struct aec_stream *strm);

/* one-shot helpers, reference src/libaec.h:165-166 (encode.c:950-963, decode.c:843-854) */
LIBAEC_API int aec_buffer_encode(struct aec_stream *strm);
LIBAEC_API int aec_buffer_decode(struct aec_stream *strm);

/*
 * Extension (no reference counterpart): n independent streams with the same parameters in ONE call --
 * the chunks of an HDF5 / netCDF dataset, where the reference's callers run one aec_buffer_* call per
 * chunk (reference src/sz_compat.c:170, 239).  Only bits_per_sample, block_size, rsi and flags of
 * `params` are read.  src[i] / src_len[i]: the i-th input; dst[i]: its output buffer, dst_len[i] its
 * capacity on entry and the bytes produced on return; status[i] (optional): AEC_OK, AEC_DATA_ERROR
 * (corrupt stream), AEC_STREAM_ERROR (encode: output did not fit, a prefix was written).  The chunks travel
 * through pinned staging (one transfer per piece of 64 MiB); decode: the RSI starts of a whole group of chunks
 * from ONE table launch (low-entropy chunks of tens of KiB and more) or one wavefront per chunk (small chunks),
 * then one decode launch per group; encode: equal chunks of whole RSIs -- the usual HDF5 case -- are analysed,
 * scanned and packed by ONE launch set for all of them (aec_gpu_encode_uniform_batch_async), other shapes by the
 * encoder kernels chunk after chunk.  Batches of more than a few MiB run as up to four parts side by side, each on
 * a HIP stream of its own.  Returns AEC_OK or the last non-OK status.
 */
LIBAEC_API int aec_buffer_encode_batch(const struct aec_stream *params, size_t n, const void *const *src,
                                       const size_t *src_len, void *const *dst, size_t *dst_len, int *status);
LIBAEC_API int aec_buffer_decode_batch(const struct aec_stream *params, size_t n, const void *const *src,
                                       const size_t *src_len, void *const *dst, size_t *dst_len, int *status);

#ifdef __cplusplus
}
#endif
#endif /* LIBAEC_H */
