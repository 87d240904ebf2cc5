/*
 * szlib.h -- SZIP-compatible entry points on top of the MI355X adaptive entropy coder.
 *
 * Same names, option masks, structure and return codes as the reference's SZIP shim
 * (reference src/szlib.h:6-43, implemented there by src/sz_compat.c), exported by
 * libaec_amd/lib/libsz.so.2, so HDF5's SZIP filter and netCDF link against it unchanged.
 */
#ifndef SZLIB_H
#define SZLIB_H 1

#include "libaec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* option masks, reference src/szlib.h:6-12 (only MSB and NN influence the coder,
 * reference src/sz_compat.c:12-27) */
#define SZ_ALLOW_K13_OPTION_MASK 1
#define SZ_CHIP_OPTION_MASK 2
#define SZ_EC_OPTION_MASK 4
#define SZ_LSB_OPTION_MASK 8
#define SZ_MSB_OPTION_MASK 16
#define SZ_NN_OPTION_MASK 32
#define SZ_RAW_OPTION_MASK 128

/* return codes, reference src/szlib.h:14-19 */
#define SZ_OK AEC_OK
#define SZ_OUTBUFF_FULL 2
#define SZ_NO_ENCODER_ERROR -1
#define SZ_PARAM_ERROR AEC_CONF_ERROR
#define SZ_MEM_ERROR AEC_MEM_ERROR

/* limits, reference src/szlib.h:21-24 */
#define SZ_MAX_PIXELS_PER_BLOCK 32
#define SZ_MAX_BLOCKS_PER_SCANLINE 128
#define SZ_MAX_PIXELS_PER_SCANLINE (SZ_MAX_BLOCKS_PER_SCANLINE) * (SZ_MAX_PIXELS_PER_BLOCK)

/* reference src/szlib.h:26-32 */
typedef struct SZ_com_t_s {
    int options_mask;
    int bits_per_pixel;       /* 1..24, or 32 / 64 (coded as byte planes) */
    int pixels_per_block;     /* block size J */
    int pixels_per_scanline;  /* one RSI per scan line, padded to whole blocks */
} SZ_com_t;

#define SZLIB_API __attribute__((visibility("default")))

/* reference src/sz_compat.c:110-183: *destLen in = capacity, out = compressed bytes;
 * SZ_OUTBUFF_FULL when the capacity is too small */
SZLIB_API int SZ_BufftoBuffCompress(void *dest, size_t *destLen, const void *source, size_t sourceLen,
                                    SZ_com_t *param);
/* reference src/sz_compat.c:185-268: *destLen in = expected bytes, out = bytes produced */
SZLIB_API int SZ_BufftoBuffDecompress(void *dest, size_t *destLen, const void *source, size_t sourceLen,
                                      SZ_com_t *param);
/*
 * Extension (no reference counterpart): n chunks of one dataset -- same SZ_com_t -- in ONE call, which is
 * how an HDF5 filter pipeline that has several chunks in hand should drive a GPU: one upload, one index +
 * one decode launch for all chunks (or the encoder kernels chunk after chunk without returning to the
 * host), one download.  Arguments as in the BufftoBuff calls, per chunk; status[i] (optional) receives
 * the per-chunk return code.  Output is byte-identical to n BufftoBuff calls.
 */
SZLIB_API int SZ_BatchCompress(void *const *dest, size_t *destLen, const void *const *source,
                               const size_t *sourceLen, size_t n, SZ_com_t *param, int *status);
SZLIB_API int SZ_BatchDecompress(void *const *dest, size_t *destLen, const void *const *source,
                                 const size_t *sourceLen, size_t n, SZ_com_t *param, int *status);
/* reference src/sz_compat.c:270-273 */
SZLIB_API int SZ_encoder_enabled(void);
/* netCDF's configure looks for this symbol (reference src/sz_compat.c:275-276) */
SZLIB_API char SZ_Compress(void);

#ifdef __cplusplus
}
#endif
#endif /* SZLIB_H */
