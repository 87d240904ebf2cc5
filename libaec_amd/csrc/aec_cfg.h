// aec_cfg.h -- host-side derivation of the coding parameters (no device code).
// Restates reference src/encode.c:777-872 (encoder validation and derived values) and
// src/decode.c:699-766 (decoder: only bits_per_sample and RESTRICTED are validated).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "aec_lane.h"

namespace aec {

enum { RC_OK = 0, RC_CONF_ERROR = -1, RC_STREAM_ERROR = -2, RC_DATA_ERROR = -3, RC_MEM_ERROR = -4 };

// Largest block the kernels handle (one lane owns one block; LDS rows are sized for it).
// The reference accepts any even size with AEC_NOT_ENFORCE but overflows its own 258-byte CDS
// buffer above 64 (reference encode.h:64-66); SZIP caps at 32 (szlib.h:21).
static const uint32_t kMaxBlockSize = 64;

// in_bytes: size of the input for encoding (trailing bytes that do not fill a sample are
// ignored, encode.c:673-698); 0 when only the derived values are wanted.
inline int make_cfg(uint32_t bps, uint32_t bs, uint32_t rsi, uint32_t flags, size_t in_bytes,
                    bool for_encode, Cfg *c)
{
    if (bps == 0 || bps > 32) return RC_CONF_ERROR;                 // encode.c:777, decode.c:699
    if (for_encode) {
        if (flags & F_NOT_ENFORCE) {
            if (bs & 1) return RC_CONF_ERROR;                       // encode.c:780-783
        } else if (bs != 8 && bs != 16 && bs != 32 && bs != 64) {
            return RC_CONF_ERROR;                                   // encode.c:785-790
        }
        if (rsi > 4096) return RC_CONF_ERROR;                       // encode.c:793
    }
    // Deliberate deviations, documented in DESIGN.md: the reference loops forever on
    // block_size == 0 / rsi == 0, invokes undefined behaviour for signed 1-bit samples
    // (shift by 32, encode.c:863) and corrupts memory for block_size > 64.
    if (bs == 0 || (bs & 1) || bs > kMaxBlockSize || rsi == 0 || rsi > 4096) return RC_CONF_ERROR;
    if ((flags & F_SIGNED) && bps == 1) return RC_CONF_ERROR;

    c->bps = bps; c->bs = bs; c->rsi = rsi; c->flags = flags;
    if (bps > 16) {
        c->id_len = 5;
        c->bytes = (bps <= 24 && (flags & F_3BYTE)) ? 3 : 4;        // encode.c:804-828
    } else if (bps > 8) {
        c->id_len = 4; c->bytes = 2;                                // encode.c:829-840
    } else {
        if (flags & F_RESTRICTED) {                                 // encode.c:843-851
            if (bps > 4) return RC_CONF_ERROR;
            c->id_len = bps <= 2 ? 1 : 2;
        } else {
            c->id_len = 3;
        }
        c->bytes = 1;
    }
    if (flags & F_SIGNED) {                                         // encode.c:862-870
        c->xmax = 0xFFFFFFFFu >> (32 - bps + 1);
        c->xmin = ~c->xmax;
    } else {
        c->xmin = 0;
        c->xmax = 0xFFFFFFFFu >> (32 - bps);
    }
    c->kmax = (1u << c->id_len) - 3u;                               // encode.c:872
    if (c->id_len == 1) c->kmax = 0;
    c->segs_per_rsi = (rsi + 63) / 64;
    c->pad0 = 0;
    c->total_samples = in_bytes / c->bytes;
    c->total_blocks = (c->total_samples + bs - 1) / bs;
    c->rsi_count = (c->total_blocks + rsi - 1) / rsi;
    const uint64_t full = c->total_blocks / rsi, rem = c->total_blocks % rsi;
    c->total_segs = full * c->segs_per_rsi + (rem + 63) / 64;
    return RC_OK;
}

// Upper bound of the encoded size in bytes: a CDS never exceeds id_len + bs*bps bits
// (uncompressed option, reference encode.c:536-545; every other option is only chosen when
// shorter, encode.c:601-611), plus up to 7 carried-in bits and word slack.
inline size_t max_encoded_bytes(const Cfg &c)
{
    return (size_t)((c.total_blocks * (uint64_t)(c.id_len + c.bs * c.bps + 2) + 7) / 8) + 16;
}

// A stream whose RSIs are as long as RSIs of uncompressed blocks (within 1/32): incompressible data.  An uncompressed
// block with a reference sample looks like one without (the reference is its first sample), so nothing in such a stream
// marks an RSI start and the schemes that guess entries from the options around a reference sample (regions, the chains
// by plausibility) only cost their passes before the trunk's block counts deliver -- 64 MiB of 8-bit noise: 22 ms of 60.
// (no RSI an encoder writes is longer than that: a hint beyond is a look-ahead, not a mean -- aec_abi.cpp passes 1.5 means
// and keeps out of this band unless the mean is in it)
inline bool index_incompressible(const Cfg &c, uint64_t rsi_bits_hint)
{
    const uint64_t raw = (uint64_t)c.rsi * (c.id_len + (uint64_t)c.bs * c.bps);
    return rsi_bits_hint + raw / 32 >= raw && rsi_bits_hint <= raw + 64;
}

}  // namespace aec
