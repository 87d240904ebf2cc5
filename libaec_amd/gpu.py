"""Device-resident batch interface (include/aec_gpu.h) for torch tensors living in HBM.

torch is used for device memory and streams only; every codec operation is a kernel launched
by libaec.so.0 on the tensor's device through the C entry points declared in aec_gpu.h.
"""
import ctypes as C

import numpy as np

from .api import library


class Params(C.Structure):
    _fields_ = [("bits_per_sample", C.c_uint), ("block_size", C.c_uint), ("rsi", C.c_uint),
                ("flags", C.c_uint)]


SEG_ENTRY_DTYPE = np.dtype([("bit", "<u8"), ("prev", "<u4"), ("pad", "<u4")])
ENC_RESULT_DTYPE = np.dtype([("total_bits", "<u8"), ("k_out", "<u4"), ("overflow", "<u4"),
                             ("k_lo", "<u4"), ("k_hi", "<u4")])
DEC_RESULT_DTYPE = np.dtype([("n_rsi", "<u8"), ("tail_blocks", "<u8"), ("end_bit", "<u8"),
                             ("status", "<u4"), ("pad", "<u4"), ("bad_rsi", "<u8")])

_bound = False


def _lib():
    global _bound
    lib = library()
    if not _bound:
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        pp = C.POINTER(Params)
        lib.aec_gpu_create.restype = C.c_int
        lib.aec_gpu_create.argtypes = [C.POINTER(vp)]
        lib.aec_gpu_destroy.restype = None
        lib.aec_gpu_destroy.argtypes = [vp]
        lib.aec_gpu_check_params.restype = C.c_int
        lib.aec_gpu_check_params.argtypes = [pp, C.c_int]
        lib.aec_gpu_encode_bound.restype = sz
        lib.aec_gpu_encode_bound.argtypes = [pp, sz]
        lib.aec_gpu_rsi_count.restype = u64
        lib.aec_gpu_rsi_count.argtypes = [pp, sz]
        lib.aec_gpu_block_count.restype = u64
        lib.aec_gpu_block_count.argtypes = [pp, sz]
        lib.aec_gpu_reserve.restype = C.c_int
        lib.aec_gpu_reserve.argtypes = [vp, pp, sz]
        lib.aec_gpu_encode_async.restype = C.c_int
        lib.aec_gpu_encode_async.argtypes = [vp, pp, vp, sz, vp, sz, C.c_uint, C.c_uint, vp, vp, vp]
        lib.aec_gpu_encode_plan_async.restype = C.c_int
        lib.aec_gpu_encode_plan_async.argtypes = [vp, pp, vp, sz, vp, vp]
        lib.aec_gpu_encode_emit_async.restype = C.c_int
        lib.aec_gpu_encode_emit_async.argtypes = [vp, pp, vp, sz, vp, sz, C.c_uint, C.c_uint, vp, vp, vp]
        lib.aec_gpu_encode_emit_planned_async.restype = C.c_int
        lib.aec_gpu_encode_emit_planned_async.argtypes = [vp, pp, vp, sz, vp, sz, vp, C.c_uint, vp, vp, vp]
        lib.aec_gpu_stitch_async.restype = C.c_int
        lib.aec_gpu_stitch_async.argtypes = [vp, sz, vp, C.c_uint, vp, sz, vp, vp]
        lib.aec_gpu_index_resume_async.restype = C.c_int
        lib.aec_gpu_index_resume_async.argtypes = [vp, pp, vp, sz, u64, C.c_uint, u64, vp, u64, vp, vp]
        lib.aec_gpu_decode_indexed_async.restype = C.c_int
        lib.aec_gpu_decode_indexed_async.argtypes = [vp, pp, vp, sz, vp, u64, vp, vp, vp, vp]
        lib.aec_gpu_decode_async.restype = C.c_int
        lib.aec_gpu_decode_async.argtypes = [vp, pp, vp, sz, vp, u64, u64, vp, vp, vp]
        lib.aec_gpu_segment_count.restype = u64
        lib.aec_gpu_segment_count.argtypes = [pp, sz]
        lib.aec_gpu_set_segment_table.restype = None
        lib.aec_gpu_set_segment_table.argtypes = [vp, vp]
        lib.aec_gpu_decode_segments_async.restype = C.c_int
        lib.aec_gpu_decode_segments_async.argtypes = [vp, pp, vp, sz, vp, u64, u64, vp, vp, vp]
        lib.aec_gpu_index_batch_async.restype = C.c_int
        lib.aec_gpu_index_batch_async.argtypes = [vp, pp, vp, sz, vp, u64, u64, vp, vp, vp]
        lib.aec_gpu_uniform_batch_ok.restype = C.c_int
        lib.aec_gpu_uniform_batch_ok.argtypes = [pp, sz, u64]
        lib.aec_gpu_encode_uniform_batch_async.restype = C.c_int
        lib.aec_gpu_encode_uniform_batch_async.argtypes = [vp, pp, vp, sz, u64, vp, sz, vp, vp, vp]
        lib.aec_gpu_index_async.restype = C.c_int
        lib.aec_gpu_index_async.argtypes = [vp, pp, vp, sz, u64, vp, u64, vp, vp]
        lib.aec_gpu_segments_per_rsi.restype = C.c_uint
        lib.aec_gpu_segments_per_rsi.argtypes = [pp]
        lib.aec_gpu_index_segments_async.restype = C.c_int
        lib.aec_gpu_index_segments_async.argtypes = [vp, pp, vp, sz, u64, C.c_uint, u64, vp, vp, u64, vp, vp]
        lib.aec_gpu_index_scheme.restype = C.c_int
        lib.aec_gpu_index_scheme.argtypes = [pp, sz, u64, C.c_uint]
        lib.aec_gpu_decode_bare_async.restype = C.c_int
        lib.aec_gpu_decode_bare_async.argtypes = [vp, pp, vp, sz, vp, vp, u64, u64, vp, vp, vp, vp]
        _bound = True
    return lib


def stitch_async(d_gathered, slot, d_plans, world, d_stream, d_total=None, stream=None):
    """aec_gpu_stitch_async: compact the all-gathered slices (world slots of `slot` bytes in d_gathered,
    16 readable bytes behind the last) into one stream at d_stream, on the device."""
    import torch
    lib = _lib()
    st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
    rc = lib.aec_gpu_stitch_async(C.c_void_p(d_gathered.data_ptr()), slot, C.c_void_p(d_plans.data_ptr()), world,
                                  C.c_void_p(d_stream.data_ptr()), d_stream.numel(),
                                  C.c_void_p(d_total.data_ptr()) if d_total is not None else None, st)
    if rc != 0:
        raise RuntimeError(f"aec_gpu_stitch_async failed ({rc})")


INDEX_SCHEMES = ("serial walk", "phase-locked chains", "window tables", "trunk", "every bit parsed (small streams)",
                 "regions walked from guessed entries (large streams)")


def index_scheme(bits_per_sample, block_size, rsi, flags, in_bytes, rsi_bits=0, start_block=0):
    """aec_gpu_index_scheme: which scheme the index pass of such a stream takes (index into INDEX_SCHEMES)."""
    p = Params(bits_per_sample, block_size, rsi, flags)
    return _lib().aec_gpu_index_scheme(C.byref(p), in_bytes, rsi_bits, start_block)


class Codec:
    """One aec_gpu context (workspace) on the current torch device."""

    def __init__(self, bits_per_sample, block_size, rsi, flags):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("libaec_amd.gpu needs a HIP device (no CPU implementation)")
        self.torch = torch
        self.lib = _lib()
        self.p = Params(bits_per_sample, block_size, rsi, flags)
        rc = self.lib.aec_gpu_check_params(C.byref(self.p), 1)
        if rc != 0:
            raise ValueError(f"invalid stream parameters (aec_gpu_check_params -> {rc})")
        self.ctx = C.c_void_p()
        torch.cuda.current_device()
        torch.zeros(1, device="cuda")          # make sure the HIP context of this device is current
        rc = self.lib.aec_gpu_create(C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_create failed ({rc})")

    def close(self):
        if self.ctx:
            self.lib.aec_gpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- sizes ---------------------------------------------------------------------------------
    def encode_bound(self, in_bytes):
        return int(self.lib.aec_gpu_encode_bound(C.byref(self.p), in_bytes))

    def rsi_count(self, in_bytes):
        return int(self.lib.aec_gpu_rsi_count(C.byref(self.p), in_bytes))

    def block_count(self, in_bytes):
        return int(self.lib.aec_gpu_block_count(C.byref(self.p), in_bytes))

    def segment_count(self, in_bytes):
        return int(self.lib.aec_gpu_segment_count(C.byref(self.p), in_bytes))

    def set_segment_table(self, d_table):
        """d_table: uint8 CUDA tensor of segment_count * 16 bytes (or None): filled by later encodes"""
        self._seg_table = d_table          # keep it alive
        self.lib.aec_gpu_set_segment_table(self.ctx, C.c_void_p(d_table.data_ptr()) if d_table is not None else None)

    def decode_segments_async(self, d_in, in_bytes, d_table, n_seg, total_blocks, d_out, d_result, stream=None):
        rc = self.lib.aec_gpu_decode_segments_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_table.data_ptr()),
            n_seg, total_blocks, C.c_void_p(d_out.data_ptr()), C.c_void_p(d_result.data_ptr()),
            self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_decode_segments_async failed ({rc})")

    def reserve(self, in_bytes):
        rc = self.lib.aec_gpu_reserve(self.ctx, C.byref(self.p), in_bytes)
        if rc != 0:
            raise MemoryError(f"aec_gpu_reserve({in_bytes}) failed ({rc})")

    def _stream(self, stream):
        return C.c_void_p(stream if stream is not None else self.torch.cuda.current_stream().cuda_stream)

    # ---- enqueue -------------------------------------------------------------------------------
    def encode_async(self, d_in, in_bytes, d_out, d_offsets, d_result, start_bit=0, k_in=0, stream=None):
        """d_in/d_out: uint8 CUDA tensors; d_offsets: int64 tensor with rsi_count+1 entries or None;
        d_result: uint8 tensor of >= 16 bytes."""
        rc = self.lib.aec_gpu_encode_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_out.data_ptr()),
            d_out.numel(), start_bit, k_in,
            C.c_void_p(d_offsets.data_ptr()) if d_offsets is not None else None,
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_encode_async failed ({rc})")

    def encode_plan_async(self, d_in, in_bytes, d_result, stream=None):
        """first half of an encode: leaves total_bits and (k_lo, k_hi) in d_result"""
        rc = self.lib.aec_gpu_encode_plan_async(self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes,
                                                C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_encode_plan_async failed ({rc})")

    def encode_emit_async(self, d_in, in_bytes, d_out, d_offsets, d_result, start_bit, k_in, stream=None):
        """second half: writes the stream at bit `start_bit` of d_out[0] with carried k `k_in`"""
        rc = self.lib.aec_gpu_encode_emit_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_out.data_ptr()),
            d_out.numel(), start_bit, k_in,
            C.c_void_p(d_offsets.data_ptr()) if d_offsets is not None else None,
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_encode_emit_async failed ({rc})")

    def encode_emit_planned_async(self, d_in, in_bytes, d_out, d_offsets, d_result, d_plans, rank, stream=None):
        """second half without a host round trip: d_plans = the all-gathered 24-byte plan records of
        all shards (uint8 tensor, world * 24 bytes), rank = this shard's index; start bit and carried
        k are computed on the device"""
        rc = self.lib.aec_gpu_encode_emit_planned_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_out.data_ptr()),
            d_out.numel(), C.c_void_p(d_plans.data_ptr()), rank,
            C.c_void_p(d_offsets.data_ptr()) if d_offsets is not None else None,
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_encode_emit_planned_async failed ({rc})")

    def index_resume_async(self, d_in, in_bytes, start_bit, start_block, rsi_start_bit, d_offsets, max_rsi,
                           d_result, stream=None):
        """d_offsets needs max_rsi + 1 entries (the last receives the start of the trailing partial RSI)"""
        rc = self.lib.aec_gpu_index_resume_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, start_bit, start_block,
            rsi_start_bit, C.c_void_p(d_offsets.data_ptr()), max_rsi, C.c_void_p(d_result.data_ptr()),
            self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_index_resume_async failed ({rc})")

    def decode_indexed_async(self, d_in, in_bytes, d_offsets, max_rsi, d_index_result, d_out, d_result, stream=None):
        rc = self.lib.aec_gpu_decode_indexed_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_offsets.data_ptr()),
            max_rsi, C.c_void_p(d_index_result.data_ptr()), C.c_void_p(d_out.data_ptr()),
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_decode_indexed_async failed ({rc})")

    def decode_async(self, d_in, in_bytes, d_offsets, n_rsi, total_blocks, d_out, d_result, stream=None):
        rc = self.lib.aec_gpu_decode_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes,
            C.c_void_p(d_offsets.data_ptr()), n_rsi, total_blocks, C.c_void_p(d_out.data_ptr()),
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_decode_async failed ({rc})")

    def index_async(self, d_in, in_bytes, start_bit, d_offsets, max_rsi, d_result, stream=None):
        rc = self.lib.aec_gpu_index_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, start_bit,
            C.c_void_p(d_offsets.data_ptr()), max_rsi, C.c_void_p(d_result.data_ptr()),
            self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_index_async failed ({rc})")

    def segments_per_rsi(self):
        return int(self.lib.aec_gpu_segments_per_rsi(C.byref(self.p)))

    def index_segments_async(self, d_in, in_bytes, start_bit, d_offsets, d_seg_bits, max_rsi, d_result, stream=None,
                             start_block=0, rsi_start_bit=0):
        """index pass that also leaves the segment starts: d_offsets int64 (max_rsi + 1), d_seg_bits int64
        ((max_rsi + 1) * segments_per_rsi())"""
        rc = self.lib.aec_gpu_index_segments_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, start_bit, start_block, rsi_start_bit,
            C.c_void_p(d_offsets.data_ptr()), C.c_void_p(d_seg_bits.data_ptr()), max_rsi,
            C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_index_segments_async failed ({rc})")

    def decode_bare_async(self, d_in, in_bytes, d_offsets, d_seg_bits, max_rsi, total_blocks, d_index_result, d_out,
                          d_result, stream=None):
        """decode behind index_segments_async, a lane per segment where the index pass found the segment starts;
        d_index_result: the index record (counts taken on the device) or None (max_rsi RSIs, total_blocks blocks)"""
        rc = self.lib.aec_gpu_decode_bare_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes, C.c_void_p(d_offsets.data_ptr()),
            C.c_void_p(d_seg_bits.data_ptr()) if d_seg_bits is not None else None, max_rsi, total_blocks,
            C.c_void_p(d_index_result.data_ptr()) if d_index_result is not None else None,
            C.c_void_p(d_out.data_ptr()), C.c_void_p(d_result.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_decode_bare_async failed ({rc})")

    def index_batch_async(self, d_in, in_bytes, d_chunk_offsets, n_chunks, rsi_per_chunk, d_offsets, d_results,
                          stream=None):
        """d_chunk_offsets: int64 tensor (n_chunks + 1 byte offsets, multiples of 16);
        d_offsets: int64 tensor (n_chunks * rsi_per_chunk); d_results: uint8 tensor (n_chunks * 40)"""
        rc = self.lib.aec_gpu_index_batch_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), in_bytes,
            C.c_void_p(d_chunk_offsets.data_ptr()), n_chunks, rsi_per_chunk, C.c_void_p(d_offsets.data_ptr()),
            C.c_void_p(d_results.data_ptr()), self._stream(stream))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_index_batch_async failed ({rc})")

    def encode_uniform_batch(self, d_in, chunk_bytes, n_chunks):
        """n equal chunks of whole RSIs, back to back in d_in, as one launch set (include/aec_gpu.h:
        aec_gpu_encode_uniform_batch_async).  Returns (d_out, records) with records[i] = (base_bits, bits) of
        stream i inside d_out; synchronises."""
        torch = self.torch
        if not self.lib.aec_gpu_uniform_batch_ok(C.byref(self.p), chunk_bytes, n_chunks):
            raise ValueError("not a uniform batch (whole RSIs, at most 2048 segments per chunk)")
        cap = (self.encode_bound(chunk_bytes) + 15) // 16 * 16 * n_chunks
        d_out = torch.empty(cap, dtype=torch.uint8, device=d_in.device)
        d_rec = torch.zeros(n_chunks * 2, dtype=torch.int64, device=d_in.device)
        d_res = torch.zeros(ENC_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=d_in.device)
        rc = self.lib.aec_gpu_encode_uniform_batch_async(
            self.ctx, C.byref(self.p), C.c_void_p(d_in.data_ptr()), chunk_bytes, n_chunks,
            C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_rec.data_ptr()), C.c_void_p(d_res.data_ptr()),
            self._stream(None))
        if rc != 0:
            raise RuntimeError(f"aec_gpu_encode_uniform_batch_async failed ({rc})")
        rec = d_rec.cpu().numpy().reshape(n_chunks, 2)
        res = d_res.cpu().numpy().view(ENC_RESULT_DTYPE)[0]
        if res["overflow"]:
            raise RuntimeError("encode overflow")
        return d_out, rec

    # ---- convenience (synchronising) -------------------------------------------------------------
    def encode(self, d_in, start_bit=0, k_in=0):
        """Encode a uint8 CUDA tensor.  Returns (d_out, n_bytes, total_bits, k_out, d_offsets)."""
        torch = self.torch
        n = d_in.numel()
        d_out = torch.empty(self.encode_bound(n), dtype=torch.uint8, device=d_in.device)
        d_off = torch.empty(self.rsi_count(n) + 1, dtype=torch.int64, device=d_in.device)
        d_res = torch.zeros(ENC_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=d_in.device)
        self.encode_async(d_in, n, d_out, d_off, d_res, start_bit, k_in)
        res = d_res.cpu().numpy().view(ENC_RESULT_DTYPE)[0]
        if res["overflow"]:
            raise RuntimeError("encode overflow")
        bits = int(res["total_bits"])
        nbytes = max(1, (start_bit + bits + 7) // 8) if (start_bit + bits) else 1
        return d_out, nbytes, bits, int(res["k_out"]), d_off

    def decode(self, d_in, in_bytes, d_offsets, n_rsi, total_blocks):
        torch = self.torch
        bps = self.p.bits_per_sample
        nb = 4 if bps > 16 and not (bps <= 24 and self.p.flags & 2) else (3 if bps > 16 else (2 if bps > 8 else 1))
        d_out = torch.empty(total_blocks * self.p.block_size * nb + 16, dtype=torch.uint8, device=d_in.device)
        d_res = torch.zeros(DEC_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=d_in.device)
        self.decode_async(d_in, in_bytes, d_offsets, n_rsi, total_blocks, d_out, d_res)
        res = d_res.cpu().numpy().view(DEC_RESULT_DTYPE)[0]
        return d_out[: total_blocks * self.p.block_size * nb], int(res["status"])
