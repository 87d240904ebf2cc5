// aec_kernels.h -- host-visible launchers of the gfx950 kernels (aec_enc.hip, aec_dec.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "aec_cfg.h"

namespace aec {

// One entry per 2048 segments for the three-phase scan of (bit length, k clamp).
struct ScanPartial {
    uint64_t bits;
    uint32_t lo, hi;
};

// Result record the encoder leaves in device memory (and the host mirrors from pinned memory).
struct EncResult {
    uint64_t total_bits;   // bits produced by this batch (excluding the carried-in start bits)
    uint32_t k_out;        // encoder's k after the last block (reference state->k)
    uint32_t overflow;     // 1 when the output capacity was too small (stores were clipped)
    uint32_t k_lo, k_hi;   // the batch's k transfer function: k_out = min(max(k_in, k_lo), k_hi)
};

// What precedes a shard of a stream that is coded on several devices (computed on the device from
// the all-gathered plan records of the preceding shards: aec_shard.hip).
struct ShardCarry {
    uint64_t start_bit;    // absolute bit position of the shard in the whole stream
    uint32_t k_in;         // carried k
    uint32_t pad;
};

struct DecResult {
    uint64_t n_rsi;        // index pass: complete RSIs found
    uint64_t tail_blocks;  // index pass: blocks of the trailing incomplete RSI
    uint64_t end_bit;      // index pass: bit after the last complete CDS
    uint32_t status;       // DEC_OK / DEC_NEED_INPUT / DEC_DATA_ERROR (worst over all lanes)
    uint32_t pad;          // index pass: 1 = the walk ended because the input did (inside a coded data set);
                           // indexed decode: samples released from that incomplete coded data set
    uint64_t bad_rsi;      // first RSI that reported a non-OK status
};

// Entry of the segment table (finer-grained sibling of the RSI offset table): where a segment
// (64 blocks) starts in the stream and the sample that precedes it, which is all a decoder needs
// to start in the middle of an RSI.
struct SegEntry {
    uint64_t bit;      // absolute start bit of the segment's first CDS
    uint32_t prev;     // raw sample just before the segment (0 for the first segment of an RSI)
    uint32_t pad;
};

struct EncWorkspace {
    uint32_t *meta;          // [total_blocks]   per-block summary (aec_lane.h meta_pack)
    uint32_t *seg_bits;      // [total_segs]     bits per segment
    uint16_t *seg_clamp;     // [total_segs]     k clamp per segment, lo | hi << 8
    uint64_t *seg_start;     // [total_segs]     absolute start bit per segment
    uint8_t *seg_kin;        // [total_segs]     k carried into the segment
    ScanPartial *partials;   // [ceil(total_segs / 2048) + 1]
    void *fused_ctl;         // single-pass encoder: ticket, fail flag, look-back granules (fused_ctl_bytes)
};

// Single-pass encoder (aec_enc.hip k_encode_fused): available for the templated block sizes.
bool fused_supported(const Cfg &c);
size_t fused_ctl_bytes(const Cfg &c);

// Optional per-phase timing: when non-null the launchers record these events around the phases.
//   encode: ev[0] | analyze | ev[1] | scan x3 | ev[2] | clear | ev[3] | pack | ev[4]
//   decode: ev[5] | decode | ev[6]
struct PhaseEvents {
    hipEvent_t ev[7];
};

static const uint32_t kScanChunk = 2048;   // segments per scan workgroup (256 threads x 8)

size_t enc_workspace_bytes(const Cfg &c, size_t *off_meta, size_t *off_bits, size_t *off_clamp,
                           size_t *off_start, size_t *off_kin, size_t *off_part);

// Enqueues analyze -> scan -> clear -> pack on `stream`.  d_out must be 4-byte aligned and hold
// out_cap bytes; bit `start_bit` (0..7) of d_out[0] is where the stream continues, k_in is the
// carried k.  d_rsi_off (optional) receives rsi_count + 1 absolute bit offsets.
//
// The work splits at the point where a batch needs to know what precedes it: PLAN (analyze + the
// local part of the scan) yields total_bits and (k_lo, k_hi) without knowing start_bit / k_in;
// EMIT (offset scan, clear, pack) needs them.  A multi-GPU single stream runs PLAN on every rank,
// exchanges the three numbers, then EMIT at the global bit offset.
enum : uint32_t { ENC_PLAN = 1, ENC_EMIT = 2, ENC_ALL = 3 };
void launch_encode(const Cfg &c, const uint8_t *d_in, uint8_t *d_out, size_t out_cap,
                   uint32_t start_bit, uint32_t k_in, const EncWorkspace &ws, uint64_t *d_rsi_off,
                   EncResult *d_res, hipStream_t stream, const PhaseEvents *prof = nullptr,
                   uint32_t phases = ENC_ALL, SegEntry *d_seg_table = nullptr,
                   const ShardCarry *d_carry = nullptr);

// A batch of n equal chunks of whole RSIs, back to back in d_in, as one launch set (aec_enc.hip): every chunk is a
// stream of its own (k = 0, zero-padded to a byte), the streams lie back to back in d_out; d_chunks[i] says where
// chunk i's stream starts and how long it is, d_res->total_bits where the last one ends (overflow: beyond out_cap).
struct BatchChunk {
    uint64_t base_bits;     // multiple of 8
    uint64_t bits;
};
bool batch_uniform_ok(const Cfg &c, uint64_t segs_per_chunk);
void launch_encode_uniform_batch(const Cfg &c, const uint8_t *d_in, uint64_t segs_per_chunk, uint8_t *d_out,
                                 size_t out_cap, const EncWorkspace &ws, BatchChunk *d_chunks, EncResult *d_res,
                                 hipStream_t stream);

// One stream over several devices (aec_shard.hip): carry-in of shard `rank` from the plan records of
// all shards; reassembly of the gathered slices into one stream.
void launch_shard_carry(const EncResult *d_plans, uint32_t rank, ShardCarry *d_carry, hipStream_t stream);
void launch_stitch(const uint8_t *d_gathered, size_t slot, const EncResult *d_plans, uint32_t world,
                   uint8_t *d_stream, size_t cap, uint64_t *d_total_bytes, hipStream_t stream);

// Enqueues the RSI-parallel decoder: one lane per RSI, offsets in bits from d_in.
//   total_blocks  blocks to produce (the last RSI may be short); d_out holds whole blocks
// d_batch (optional, with rsi_per_chunk): records of launch_index_batch, one per independent stream; stream
// s owns RSIs [s * rsi_per_chunk, ...) of the table and of the output, and its record says how many hold blocks.
// d_idx (optional): the record an index pass left on the device; the kernel then takes the number of
// RSIs and blocks from it (n_rsi = the most the index pass could find, it sizes the grid).
// false = the device-side scratch the kernels need could not be allocated.
bool launch_decode(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_rsi_off,
                   uint64_t n_rsi, uint64_t total_blocks, uint8_t *d_out, DecResult *d_res,
                   hipStream_t stream, const PhaseEvents *prof = nullptr, const DecResult *d_idx = nullptr,
                   const DecResult *d_batch = nullptr, uint32_t rsi_per_chunk = 0);

// Samples of the coded data set the input ends in (single lane; see k_decode_partial): after
// launch_decode with the same d_idx / d_out / d_res; their number is left in d_res->pad.
void launch_decode_partial(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const DecResult *d_idx,
                           uint8_t *d_out, DecResult *d_res, hipStream_t stream);

// Same, one lane per SEGMENT (needs the encoder's segment table): the way to fill the chip when
// RSIs are large (32-bit, block 32, rsi 4096 = 512 KiB per RSI, only 8192 RSIs in 4 GiB).
bool launch_decode_segments(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const SegEntry *d_seg_table,
                            uint64_t n_seg, uint64_t total_blocks, uint8_t *d_out, DecResult *d_res,
                            hipStream_t stream, const PhaseEvents *prof = nullptr);

// A bare stream segment by segment: like launch_decode with the segment starts an index pass found beside the RSI
// starts (launch_index with d_seg_bits: entry [r * segs_per_rsi + j], ~0 = unknown) -- a summing pass gives every
// segment the sample in front of it (inside the range the inverse predictor is a running sum), then one lane per
// segment; RSIs that cannot be taken that way are decoded by one lane each.  Falls back to launch_decode where
// decode_bare_supported() says no or the workspace (decode_bare_workspace_bytes) is missing.
// avg_cds_hint: bits per coded data set (0 = unknown), sizes rings and loads in flight when d_idx is given.
bool decode_bare_supported(const Cfg &c);
size_t decode_bare_workspace_bytes(const Cfg &c, uint64_t max_rsi);
bool launch_decode_bare(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_rsi_off,
                        const uint64_t *d_seg_bits, uint64_t n_rsi, uint64_t total_blocks, uint8_t *d_out,
                        DecResult *d_res, hipStream_t stream, const PhaseEvents *prof, const DecResult *d_idx,
                        void *d_ws, size_t ws_bytes, uint64_t avg_cds_hint);

// Enqueues the RSI index pass over one stream starting at start_bit (an RSI boundary).  With a
// workspace (index_workspace_bytes() says how much is wanted, 0 = this input takes the serial walk alone;
// less than that means more, smaller spans) the walk hops over the trunk tables built by all CUs (aec_idx.hip).
// rsi_bits_hint: estimate of the coded size of one RSI (0 = unknown); it sizes burn-in, regions and records.
size_t index_workspace_bytes(const Cfg &c, size_t in_bytes, uint64_t start_bit, uint64_t rsi_bits_hint);
// ... without the every-bit scheme's tables (8 .. 16 bytes per bit of a small stream): what to ask for when the full
// size cannot be had -- launch_index takes whatever scheme the workspace it is given has room for
size_t index_workspace_bytes_large(const Cfg &c, size_t in_bytes, uint64_t start_bit, uint64_t rsi_bits_hint);
bool index_is_windowed(const Cfg &c, size_t in_bytes, uint64_t rsi_bits_hint);
// the scheme launch_index takes with the workspace index_workspace_bytes asks for: 0 serial walk, 1 phase-locked
// chains, 2 window tables, 3 trunk, 4 every bit parsed (small streams), 5 regions walked from guessed entries (large
// streams; what it does not deliver is left to the scheme this function would name without it).  It describes the plain
// index pass (aec_gpu_index_async / aec_gpu_index_resume_async): with segment starts (d_seg_bits) or as a piece of a
// longer stream (stop_near) launch_index skips schemes 4 and 1, which deliver neither.
int index_scheme(const Cfg &c, size_t in_bytes, uint64_t rsi_bits_hint, uint32_t start_block);
// d_seg_bits (optional; (max_rsi + 1) * segs_per_rsi entries, set to ~0 by the caller): where the index runs over
// the trunk tables it also leaves the start bit of every segment of the RSIs it finds (launch_decode_bare); the
// return value says whether it did.
// stop_near (bits, window-table index only): the caller holds more of the stream than it hands in; an RSI that the
// tables leave unresolved within that many bits of the end of the input is left to the caller's next piece (the pass
// ends in front of it) instead of being walked serially.
bool launch_index(const Cfg &c, const uint8_t *d_in, size_t in_bytes, uint64_t start_bit,
                  uint64_t *d_rsi_off, uint64_t max_rsi, DecResult *d_res, hipStream_t stream,
                  void *d_ws = nullptr, size_t ws_bytes = 0, uint64_t rsi_bits_hint = 0,
                  uint32_t start_block = 0, uint64_t rsi_start = 0, uint32_t tail_slot = 0,
                  uint64_t *d_seg_bits = nullptr, uint64_t stop_near = 0);

// The region index (aec_region.hip): large preprocessed streams -- a lane per region guesses the first RSI start behind
// the region's first bit from the options around it, lanes walk the regions from their entries, every entry is checked
// against the walk in front and mended; delivers only if all agree.  Returns the device flag that is != 0 once the
// stream has been delivered (the skip_if of the schemes enqueued behind).
struct RegionPlan {
    bool ok;
    uint32_t nreg, budget, passes, avg_cds, K;
    uint64_t region_bits;
    size_t o_flags, o_found, o_entry[2], o_exit[2], o_cnt[2], o_base, o_list, o_slist, bytes;
};
RegionPlan region_plan(const Cfg &c, uint64_t total_bits, uint64_t rsi_bits_hint, bool want_segments);
const uint32_t *launch_index_regions(const Cfg &c, const RegionPlan &p, const uint32_t *words, uint64_t nwords,
                                     uint64_t end_bit, uint64_t start_bit, uint64_t *d_rsi_off, uint64_t max_rsi,
                                     DecResult *d_res, hipStream_t st, uint8_t *base, uint32_t start_block, uint64_t rsi_start,
                                     uint32_t tail_slot, uint64_t *d_seg_bits, const uint32_t *skip_if = nullptr);

// Index pass over many independent streams stored in one buffer (e.g. the chunks of an HDF5
// dataset): stream s occupies bytes [chunk_off[s], chunk_off[s+1]) (chunk_off 4-byte aligned values,
// n_chunks + 1 entries); one wavefront per stream, rsi_per_chunk offsets per stream, absolute bit
// positions, so the table can go straight into launch_decode for the whole batch.
void launch_index_batch(const Cfg &c, const uint8_t *d_in, size_t in_bytes, const uint64_t *d_chunk_off,
                        uint64_t n_chunks, uint64_t rsi_per_chunk, uint64_t *d_rsi_off, DecResult *d_res,
                        hipStream_t st, void *d_ws = nullptr, size_t ws_bytes = 0, size_t max_chunk_bytes = 0,
                        uint64_t rsi_bits_hint = 0);
// workspace with which the batch walk hops over window tables built in ONE launch for all streams (0 = the streams
// are not the low-entropy kind, or too much for one span of tables: every stream is walked serially)
size_t index_batch_workspace_bytes(const Cfg &c, size_t in_bytes, uint64_t n_chunks, size_t max_chunk_bytes,
                                   uint64_t rsi_bits_hint);

}  // namespace aec
