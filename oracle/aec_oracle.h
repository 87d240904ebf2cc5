/*
 * aec_oracle.h -- CPU restatement of the CCSDS 121.0-B-2 adaptive entropy coder as
 * implemented by erget/libaec 0.3.4 (reference tree: /root/reference).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker.  The product path (libaec_amd/) never links or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   - the reference's shipped known-answer file data/typical.rz (decode and
 *     byte-identical re-encode, -n16 -j64 -r256 -m + preprocess; src/benc.sh:7),
 *   - the option-ID assertions and data patterns of tests/check_code_options.c,
 *     tests/check_buffer_sizes.c and tests/check_long_fs.c,
 *   - golden vectors under tests/golden/ produced by the reference itself
 *     (oracle/_ref, compiled from /root/reference by oracle/Makefile) with the
 *     committed generator tests/golden/make_golden.py,
 *   - and, whenever oracle/_ref is present, randomized differential sweeps.
 */
#ifndef AEC_ORACLE_H
#define AEC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag bits: same values as the public ABI (reference src/libaec.h:105-124) */
#define AECO_SIGNED      1
#define AECO_3BYTE       2
#define AECO_MSB         4
#define AECO_PREPROCESS  8
#define AECO_RESTRICTED  16
#define AECO_PAD_RSI     32
#define AECO_NOT_ENFORCE 64

/* return codes: same values as reference src/libaec.h:129-133 */
#define AECO_OK            0
#define AECO_CONF_ERROR   (-1)
#define AECO_STREAM_ERROR (-2)
#define AECO_DATA_ERROR   (-3)
#define AECO_MEM_ERROR    (-4)

/* code option recorded in the per-block trace */
enum { AECO_OPT_ZERO = 0, AECO_OPT_SE = 1, AECO_OPT_SPLIT = 2, AECO_OPT_UNCOMP = 3,
       AECO_OPT_ZERO_CONT = 4 /* block swallowed by a zero run started earlier */ };

typedef struct aeco_params {
    unsigned bits_per_sample;
    unsigned block_size;
    unsigned rsi;
    unsigned flags;
} aeco_params;

/* One record per input block, in stream order (optional output of aeco_encode). */
typedef struct aeco_block_trace {
    uint8_t  option;   /* AECO_OPT_* */
    uint8_t  k;        /* encoder's state->k after this block (reference encode.c:407) */
    uint16_t reserved;
    uint32_t bits;     /* bits this block contributed to the stream (0 for ZERO_CONT) */
} aeco_block_trace;

/*
 * Whole-buffer encode, equivalent to reference aec_buffer_encode (encode.c:950-963):
 * init, one aec_encode(AEC_FLUSH), end.  Trailing bytes that do not make a whole
 * sample are ignored (encode.c:673-698).
 *
 *   out_len      bytes written (max(1, ceil(total_bits/8)), encode.c:686-695)
 *   trace        NULL or array of >= ceil(nsamples/block_size) records
 *   rsi_bit_off  NULL or array of >= ceil(nblocks/rsi) entries: start bit of every RSI
 *   total_bits   NULL or receives number of stream bits before final padding
 * returns AECO_OK, AECO_CONF_ERROR, or AECO_STREAM_ERROR when out_cap is too small
 * (what aec_encode_end reports, encode.c:944-945).
 */
int aeco_encode(const aeco_params *p, const uint8_t *in, size_t in_len,
                uint8_t *out, size_t out_cap, size_t *out_len,
                aeco_block_trace *trace, uint64_t *rsi_bit_off, uint64_t *total_bits);

/*
 * Whole-buffer decode, equivalent to reference aec_buffer_decode (decode.c:843-854).
 * Decodes until the input is exhausted or the output is full, with the sample-granular
 * stopping rules of the reference's resumable states (decode.c:342-400, 423-460,
 * 504-516, 560-587, 646-657).
 *   out_len   bytes written
 *   in_used   NULL or receives the number of input bits consumed by completed CDSes
 * returns AECO_OK, AECO_CONF_ERROR, AECO_DATA_ERROR (zero-run overrun, decode.c:543-544)
 * or AECO_MEM_ERROR (0 < remaining output < one sample, decode.c:821-823).
 */
int aeco_decode(const aeco_params *p, const uint8_t *in, size_t in_len,
                uint8_t *out, size_t out_cap, size_t *out_len, uint64_t *in_used);

/* Derived coding parameters (reference encode.c:804-872, decode.c:711-766). */
typedef struct aeco_derived {
    int id_len;
    int bytes_per_sample;
    int kmax;
    uint32_t xmin, xmax;
} aeco_derived;
int aeco_derive(const aeco_params *p, int for_encode, aeco_derived *d);

#ifdef __cplusplus
}
#endif
#endif
