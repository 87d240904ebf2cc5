/*
 * aec_gpu.h -- device-resident batch interface of the MI355X adaptive entropy coder.
 *
 * The libaec ABI (libaec.h) hands over host buffers; these entry points are what sits directly
 * under it and what a GPU-resident caller (an HDF5 VOL/filter that keeps chunks in HBM, the
 * benchmark, the multi-GPU driver) binds instead: all pointers marked d_ are HIP device
 * pointers, every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and
 * returns; results are written to device memory.  No reference counterpart: the reference is
 * CPU-only (this replaces the per-RSI loops of reference src/encode.c:709-754 and
 * src/decode.c:797-831 for a whole batch of RSIs at once).
 */
#ifndef AEC_GPU_H
#define AEC_GPU_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEC_GPU_API __attribute__((visibility("default")))

typedef struct aec_gpu_ctx aec_gpu_ctx;

/* same four values a caller puts into struct aec_stream (reference src/libaec.h:84-94) */
typedef struct aec_gpu_params {
    unsigned int bits_per_sample;
    unsigned int block_size;
    unsigned int rsi;
    unsigned int flags;
} aec_gpu_params;

typedef struct aec_gpu_enc_result {
    uint64_t total_bits;  /* stream bits produced by the call (without the carried-in start bits) */
    uint32_t k_out;       /* encoder's carried k after the last block (reference state->k) */
    uint32_t overflow;    /* 1: out_cap was too small, output clipped */
    uint32_t k_lo, k_hi;  /* the call's k transfer function: k_out = min(max(k_in, k_lo), k_hi) */
} aec_gpu_enc_result;

typedef struct aec_gpu_dec_result {
    uint64_t n_rsi;        /* index pass: complete RSIs found */
    uint64_t tail_blocks;  /* index pass: complete blocks of the trailing partial RSI; decode records: the lowest
                              failing block of the batch (numbered from the first RSI), ~0 if none -- every block
                              in front of it is decoded and valid */
    uint64_t end_bit;      /* index pass: bit position after the last complete coded data set */
    uint32_t status;       /* 0 ok, 1 input ended inside a coded data set, 2 corrupt stream */
    uint32_t pad;          /* index pass: 1 = stopped because the input ended; aec_gpu_decode_indexed_async:
                              samples released from the coded data set the input ends in (bits 0..30).
                              Decode records: bit 31 = a coded data set longer than any libaec's encoder
                              writes was met and the batch went through the sequential decoder (informational:
                              the output is complete and exact either way) */
    uint64_t bad_rsi;      /* lowest RSI with status != 0 */
} aec_gpu_dec_result;

/* Entry of the segment table: a segment = 64 consecutive blocks of one RSI (the last segment of
 * an RSI may be shorter).  Start bit of its first coded data set and the raw sample preceding it
 * are all a decoder needs to start there, so decoding parallelises over segments instead of RSIs:
 * the difference between 8192 and 524288 work items for 4 GiB of 32-bit / block 32 / rsi 4096. */
typedef struct aec_gpu_seg_entry {
    uint64_t bit;
    uint32_t prev;
    uint32_t pad;
} aec_gpu_seg_entry;

/* Context = workspace on the current HIP device.  One context per host thread / stream. */
AEC_GPU_API int aec_gpu_create(aec_gpu_ctx **ctx);
AEC_GPU_API void aec_gpu_destroy(aec_gpu_ctx *ctx);

/* AEC_OK or AEC_CONF_ERROR for a parameter set (same rules as aec_encode_init / aec_decode_init) */
AEC_GPU_API int aec_gpu_check_params(const aec_gpu_params *p, int for_encode);

/* Output capacity (bytes, multiple of 16) that can never overflow for in_bytes of input. */
AEC_GPU_API size_t aec_gpu_encode_bound(const aec_gpu_params *p, size_t in_bytes);
/* Number of RSIs / blocks in_bytes of input make (entries in the offset table = n_rsi + 1). */
AEC_GPU_API uint64_t aec_gpu_rsi_count(const aec_gpu_params *p, size_t in_bytes);
AEC_GPU_API uint64_t aec_gpu_block_count(const aec_gpu_params *p, size_t in_bytes);

/* Grow the context's workspace for inputs up to in_bytes now (allocation synchronises the
 * device; doing it here keeps the enqueue calls below free of allocations). */
AEC_GPU_API int aec_gpu_reserve(aec_gpu_ctx *ctx, const aec_gpu_params *p, size_t in_bytes);

/*
 * Encode in_bytes at d_in (16-byte aligned) into d_out (16-byte aligned, out_cap a multiple of
 * 16).  The stream continues at bit `start_bit` (0..7) of d_out[0] with carried k `k_in` (both
 * 0 for a new stream); bits before start_bit are left zero for the caller to merge.  All whole
 * samples are coded; a final partial block is padded as aec_encode(AEC_FLUSH) does.  The final
 * zero padding to a byte boundary is implicit (the buffer is cleared up to the last word).
 * d_rsi_bit_offsets (optional) receives rsi_count + 1 absolute bit positions: the start of every
 * RSI and the end of the stream.  d_result receives an aec_gpu_enc_result.
 */
AEC_GPU_API int aec_gpu_encode_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                     size_t in_bytes, void *d_out, size_t out_cap,
                                     unsigned int start_bit, unsigned int k_in,
                                     uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result,
                                     void *stream);

/*
 * The same work in two steps for callers that code ONE stream on several devices: PLAN analyses
 * the input and leaves total_bits and (k_lo, k_hi) in d_result without needing start_bit / k_in;
 * after the caller has combined the plans of all preceding parts (sum of total_bits; composition of
 * the k clamps) EMIT writes the part at its global bit offset.  Both calls must use the same
 * context, parameters and input, with no other encode on that context in between.
 */
AEC_GPU_API int aec_gpu_encode_plan_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                          size_t in_bytes, aec_gpu_enc_result *d_result, void *stream);
AEC_GPU_API int aec_gpu_encode_emit_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                          size_t in_bytes, void *d_out, size_t out_cap,
                                          unsigned int start_bit, unsigned int k_in,
                                          uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result,
                                          void *stream);

/*
 * The same without a host round trip between PLAN and EMIT: d_plans is the array of the plan records
 * (aec_gpu_enc_result, as written by aec_gpu_encode_plan_async) of ALL shards in stream order -- with
 * RCCL: ncclAllGather(sendbuff = the local record, recvbuff = d_plans, sendcount = 24, ncclUint8) --
 * and `rank` this shard's index.  The start bit (sum of the preceding total_bits) and the carried k
 * (composition of the preceding clamps) are computed on the device; the shard is written at bit
 * (start % 8) of d_out[0].  aec_gpu_stitch_async reassembles the all-gathered slices
 * (ncclAllGather of d_out[0 .. slot_bytes) of every rank into d_gathered; slot_bytes a multiple of 16
 * that holds the largest slice; 16 readable bytes behind the last slot) into ONE stream at d_stream:
 * slice r lands at byte (start_r / 8), bytes shared by two slices are OR-ed.  *d_total_bytes
 * (optional) receives the stream length, max(1, ceil(total bits / 8)).  At most 64 shards.
 */
AEC_GPU_API int aec_gpu_encode_emit_planned_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                                  size_t in_bytes, void *d_out, size_t out_cap,
                                                  const aec_gpu_enc_result *d_plans, unsigned int rank,
                                                  uint64_t *d_rsi_bit_offsets, aec_gpu_enc_result *d_result,
                                                  void *stream);
AEC_GPU_API int aec_gpu_stitch_async(const void *d_gathered, size_t slot_bytes,
                                     const aec_gpu_enc_result *d_plans, unsigned int world, void *d_stream,
                                     size_t stream_cap, uint64_t *d_total_bytes, void *stream);

/*
 * Decode n_rsi RSIs whose start bits are d_rsi_bit_offsets[0..n_rsi) (relative to d_in, which
 * must be 4-byte aligned and readable up to the next multiple of 4) into d_out, producing
 * `total_blocks` whole blocks (the last RSI may be short).  d_out needs
 * total_blocks * block_size * bytes_per_sample bytes and 16-byte alignment.
 */
AEC_GPU_API int aec_gpu_decode_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                     size_t in_bytes, const uint64_t *d_rsi_bit_offsets,
                                     uint64_t n_rsi, uint64_t total_blocks, void *d_out,
                                     aec_gpu_dec_result *d_result, void *stream);

/*
 * Segment-granular variant of the offset table.  After aec_gpu_set_segment_table(ctx, d_table) every
 * encode / emit call on ctx also fills d_table with aec_gpu_segment_count() entries (pass NULL to
 * stop).  aec_gpu_decode_segments_async decodes from such a table, one lane per segment.
 */
AEC_GPU_API uint64_t aec_gpu_segment_count(const aec_gpu_params *p, size_t in_bytes);
AEC_GPU_API void aec_gpu_set_segment_table(aec_gpu_ctx *ctx, aec_gpu_seg_entry *d_table);
AEC_GPU_API int aec_gpu_decode_segments_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                              size_t in_bytes, const aec_gpu_seg_entry *d_seg_table,
                                              uint64_t n_seg, uint64_t total_blocks, void *d_out,
                                              aec_gpu_dec_result *d_result, void *stream);

/*
 * Find the RSI start offsets of a stream that comes without an offset table, from bit start_bit
 * (an RSI boundary) to the end of the input, to max_rsi RSIs, or to a corrupt coded data set.
 * d_rsi_bit_offsets needs max_rsi entries.  The reference has no counterpart (its decoder walks
 * the stream CDS by CDS, src/decode.c:402-421).  Here all CUs first tabulate, for every bit
 * position, the length of an RSI that would start there (speculation through an LDS window:
 * rank/select over the 1-bits gives the end of a coded data set in O(1)); one wavefront then hops
 * from RSI start to RSI start over those tables.  RSIs longer than the look-ahead of a window
 * (about 20 kbit coded), cut by the end of the input or malformed are walked serially, so the
 * result never depends on the tables.  The look-ahead is sized from the expected coded RSI:
 * (input bits / max_rsi) unless aec_gpu_set_index_hint gave a better estimate (0 = back to default).
 * The call may allocate table workspace in the context (aec_gpu_held_bytes tells how much: 0.4 GB for 64 MiB, 2.3 GB for
 * 4 GiB of 16-bit data, up to 4.5 GB for 4 GiB of 32-bit data; larger inputs are indexed span by span within that).
 */
AEC_GPU_API void aec_gpu_set_index_hint(aec_gpu_ctx *ctx, uint64_t rsi_bits);
/* 1 when the index pass of a stream of in_bytes whose coded RSIs average rsi_bits runs over the window tables
 * (low-entropy streams, RSIs inside a 64-kbit window): cheap per call also for pieces of a few MiB, so a caller
 * may decode such a stream piece by piece and overlap one piece's transfers with the next one's kernels. */
AEC_GPU_API int aec_gpu_index_is_windowed(const aec_gpu_params *p, size_t in_bytes, uint64_t rsi_bits);
/* Which scheme the index pass of such a stream takes FIRST, given the workspace it asks for (aec_gpu_index_async /
 * aec_gpu_index_resume_async; with segment starts or as a piece of a longer stream schemes 4 and 1 are skipped):
 * 0 = the serial walk alone, 1 = phase-locked chains (RSIs of at most 44 blocks; entries by plausibility for long coded
 * data sets), 2 = window tables, 3 = the trunk, 4 = every bit parsed and the RSI starts by pointer doubling (streams of
 * at most 2 MiB with RSIs of at most 64 blocks or that the window tables do not serve, chunks of at most 128 KiB with RSIs
 * of up to 256 blocks, AEC_PAD_RSI and walks that resume inside an RSI included; without the preprocessor streams of any
 * size with RSIs of at most 256 blocks, piece by piece; with it up to 4 MiB where entries would be guessed by
 * plausibility), 5 = regions walked from guessed entries (large preprocessed streams).  A scheme that gives a stream up
 * on the device (guesses judged wrong, tables that resolve too little) is followed by the next one enqueued behind it --
 * scheme 4 behind 1 and 2 for streams of moderate size, then the serial walk -- which this function does not tell.
 * start_block != 0: a walk that resumes inside an RSI.  Host arithmetic only; tests assert the path instead of a time. */
AEC_GPU_API int aec_gpu_index_scheme(const aec_gpu_params *p, size_t in_bytes, uint64_t rsi_bits, unsigned int start_block);
/* The NEXT index pass on ctx (one pass only) is handed a piece of a stream of which the caller holds more: an RSI
 * that the window tables leave unresolved within stop_near_bits of the end of the piece -- they end there for lack
 * of look-ahead -- is not walked serially; the pass ends in front of it (n_rsi RSIs, tail_blocks 0, end_bit = its
 * start) and the caller's next piece starts there. */
AEC_GPU_API void aec_gpu_set_index_piece(aec_gpu_ctx *ctx, uint64_t stop_near_bits);

/* Release the context's workspaces that are larger than keep_bytes (they are re-allocated on
 * demand); for callers that keep a context around between jobs of very different size. */
AEC_GPU_API void aec_gpu_trim(aec_gpu_ctx *ctx, size_t keep_bytes);
/* Device memory the context holds for its own purposes at the moment (encoder workspace, index tables). */
AEC_GPU_API size_t aec_gpu_held_bytes(const aec_gpu_ctx *ctx);
AEC_GPU_API int aec_gpu_index_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                    size_t in_bytes, uint64_t start_bit, uint64_t *d_rsi_bit_offsets,
                                    uint64_t max_rsi, aec_gpu_dec_result *d_result, void *stream);

/*
 * Streaming callers (the libaec ABI layer): continue an index pass INSIDE an RSI.  start_bit is a
 * coded-data-set boundary, `start_block` blocks of the current RSI (which began at rsi_start_bit) lie
 * before it; entry 0 of the offset table receives rsi_start_bit, n_rsi / tail_blocks count from that
 * RSI.  The table needs max_rsi + 1 entries here: entry [max_rsi] receives the start bit of the RSI the
 * pass ended in (the trailing partial RSI when tail_blocks != 0).  aec_gpu_decode_indexed_async then decodes what such a pass found WITHOUT the host reading its
 * result first: the kernel takes the counts from d_index_result on the device; max_rsi (the bound given
 * to the index pass) sizes the launch, d_out must hold max_rsi whole RSIs and one block.  As the reference's
 * resumable readers do (src/decode.c:423-460), it also releases the samples of the coded data set the input
 * ends in whose bits have arrived: d_result->pad of them, behind the last complete block.
 * d_result must be a different record than d_index_result.
 */
AEC_GPU_API int aec_gpu_index_resume_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                           size_t in_bytes, uint64_t start_bit, unsigned int start_block,
                                           uint64_t rsi_start_bit, uint64_t *d_rsi_bit_offsets,
                                           uint64_t max_rsi, aec_gpu_dec_result *d_result, void *stream);
AEC_GPU_API int aec_gpu_decode_indexed_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                             size_t in_bytes, const uint64_t *d_rsi_bit_offsets,
                                             uint64_t max_rsi, const aec_gpu_dec_result *d_index_result,
                                             void *d_out, aec_gpu_dec_result *d_result, void *stream);

/*
 * Bare streams with LONG RSIs (32-bit / block 32 / rsi 4096: 512 KiB per RSI, 8192 RSIs in 4 GiB): a lane per RSI
 * leaves the device idle, and the reference's format has no entry points inside an RSI (src/decode.c:402-421).
 * aec_gpu_index_segments_async is aec_gpu_index_resume_async (start_block 0: a plain index pass from an RSI start)
 * that also leaves the start bit of every SEGMENT of 64 blocks of the RSIs it finds:
 * d_seg_bits[r * aec_gpu_segments_per_rsi() + j], (max_rsi + 1) * aec_gpu_segments_per_rsi() entries, ~0 where it
 * does not know one (streams it does not index over the trunk tables: all of them).  aec_gpu_decode_bare_async
 * then decodes with one lane per segment: a first pass sums the predictor's steps per segment (inside the sample
 * range the inverse predictor of src/decode.c:96-134 is a running sum; where a sample comes within reach of the
 * range's ends the RSI is decoded by one lane as before), which gives every segment the sample in front of it,
 * the second pass decodes.  Counts: from d_index_result on the device when given (as aec_gpu_decode_indexed_async,
 * incl. the samples of the coded data set the input ends in), else max_rsi RSIs holding total_blocks blocks.
 * Both calls on the same context, the decode behind the index pass whose tables it takes.
 */
AEC_GPU_API unsigned int aec_gpu_segments_per_rsi(const aec_gpu_params *p);
AEC_GPU_API int aec_gpu_index_segments_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                             size_t in_bytes, uint64_t start_bit, unsigned int start_block,
                                             uint64_t rsi_start_bit, uint64_t *d_rsi_bit_offsets,
                                             uint64_t *d_seg_bits, uint64_t max_rsi, aec_gpu_dec_result *d_result,
                                             void *stream);
AEC_GPU_API int aec_gpu_decode_bare_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                          size_t in_bytes, const uint64_t *d_rsi_bit_offsets,
                                          const uint64_t *d_seg_bits, uint64_t max_rsi, uint64_t total_blocks,
                                          const aec_gpu_dec_result *d_index_result, void *d_out,
                                          aec_gpu_dec_result *d_result, void *stream);

/*
 * Many independent streams in one launch -- the shape of an HDF5 / netCDF dataset stored as SZIP
 * chunks, where the parallelism of decoding comes from the chunks.  Stream s occupies bytes
 * [d_chunk_offsets[s], d_chunk_offsets[s+1]) of d_in (n_chunks + 1 entries, every offset a multiple
 * of 16) and decodes to exactly rsi_per_chunk RSIs.  One wavefront walks each stream and writes its
 * RSI start bits (absolute in d_in) to d_rsi_bit_offsets[s * rsi_per_chunk ..] and a result record
 * to d_results[s] (n_rsi < rsi_per_chunk: the stream was shorter than announced; status 2: corrupt).
 * The filled table then decodes the whole batch with ONE aec_gpu_decode_async call
 * (n_rsi = n_chunks * rsi_per_chunk), chunk s landing at RSI s * rsi_per_chunk of d_out.
 */
AEC_GPU_API int aec_gpu_index_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                          size_t in_bytes, const uint64_t *d_chunk_offsets,
                                          uint64_t n_chunks, uint64_t rsi_per_chunk,
                                          uint64_t *d_rsi_bit_offsets, aec_gpu_dec_result *d_results,
                                          void *stream);

/*
 * The two steps above in one enqueue, also for streams whose last RSI is short: index pass (one wavefront
 * per stream) + ONE decode launch that takes every stream's RSI / block counts from its record in
 * d_results.  Stream s decodes to d_out + s * rsi_per_chunk * (rsi * block_size * bytes per sample); the
 * blocks it produced: d_results[s].n_rsi * rsi + d_results[s].tail_blocks.  d_result: overall decode
 * status (worst over all lanes).  This is the shape of reading an HDF5 / netCDF dataset of SZIP chunks
 * (the call site it serves: one aec_buffer_decode per chunk, reference src/sz_compat.c:239).
 */
AEC_GPU_API int aec_gpu_decode_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                           size_t in_bytes, const uint64_t *d_chunk_offsets, uint64_t n_chunks,
                                           uint64_t rsi_per_chunk, uint64_t *d_rsi_bit_offsets, void *d_out,
                                           aec_gpu_dec_result *d_results, aec_gpu_dec_result *d_result,
                                           void *stream);

/* 1 when aec_gpu_decode_batch_async would find the RSI starts of such a batch (n_chunks streams in in_bytes,
 * rsi_per_chunk RSIs each) over window tables built in ONE launch for all streams -- low-entropy streams of tens of
 * KiB and more, at most about 16 MiB of them per call -- instead of walking every stream serially. */
AEC_GPU_API int aec_gpu_batch_uses_tables(aec_gpu_ctx *ctx, const aec_gpu_params *p, size_t in_bytes,
                                          uint64_t n_chunks, uint64_t rsi_per_chunk);

/*
 * n independent streams coded from one device buffer: chunk i = bytes [chunk_offsets[i],
 * chunk_offsets[i+1]) of d_in (HOST array of n_chunks + 1 offsets, each a multiple of 16) is coded as a
 * stream of its own (k = 0, bit 0) into d_out + i * slot_bytes (slot_bytes a multiple of 16, at least
 * aec_gpu_encode_bound of the largest chunk); d_results[i] receives its total_bits.  Everything is
 * enqueued on `stream`, nothing returns to the host (reference call site: one aec_buffer_encode per
 * chunk, src/sz_compat.c:170).
 */
AEC_GPU_API int aec_gpu_encode_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                           const uint64_t *chunk_offsets, uint64_t n_chunks, void *d_out,
                                           size_t slot_bytes, aec_gpu_enc_result *d_results, void *stream);

/*
 * The same for n_chunks EQUAL chunks of whole RSIs lying back to back in d_in (chunk i = bytes [i * chunk_bytes,
 * (i + 1) * chunk_bytes)) -- HDF5's chunks through the SZIP filter (src/sz_compat.c:170) -- as ONE launch set for
 * the whole batch instead of one per chunk: the streams come out back to back in d_out, each zero-padded to a
 * byte as aec_buffer_encode pads it; d_chunks[i] = where stream i starts (bits, a multiple of 8) and its length
 * in bits, d_result->total_bits = the end of the last one, d_result->overflow = 1 if that lies beyond out_cap
 * (the sum of aec_gpu_encode_bound(chunk_bytes) always suffices).  aec_gpu_uniform_batch_ok says whether a
 * geometry can be taken this way (whole RSIs, at most 2048 segments of 64 blocks per chunk); if not, or for
 * unequal chunks, use aec_gpu_encode_batch_async.
 */
typedef struct aec_gpu_batch_chunk {
    uint64_t base_bits;
    uint64_t bits;
} aec_gpu_batch_chunk;
AEC_GPU_API int aec_gpu_uniform_batch_ok(const aec_gpu_params *p, size_t chunk_bytes, uint64_t n_chunks);
AEC_GPU_API int aec_gpu_encode_uniform_batch_async(aec_gpu_ctx *ctx, const aec_gpu_params *p, const void *d_in,
                                                   size_t chunk_bytes, uint64_t n_chunks, void *d_out, size_t out_cap,
                                                   aec_gpu_batch_chunk *d_chunks, aec_gpu_enc_result *d_result,
                                                   void *stream);

/*
 * Measurement hooks (bench.py): with profiling enabled the context records HIP events on the
 * caller's stream around its kernels, one event set per call (a ring of 32), so a timed loop
 * needs no synchronisation inside it; aec_gpu_phase_ms waits for them and returns the device
 * time per phase (analyze, scan, clear, pack of the encode calls; decode) in milliseconds,
 * averaged over the calls made since profiling was enabled (the last 32 at most; -1 for a
 * phase that has not run).  Enabling profiling again restarts the average.
 */
AEC_GPU_API int aec_gpu_profile(aec_gpu_ctx *ctx, int enable);
AEC_GPU_API int aec_gpu_phase_ms(aec_gpu_ctx *ctx, float ms[5]);

#ifdef __cplusplus
}
#endif
#endif /* AEC_GPU_H */
