"""ctypes mirror of the SZIP entry points (include/szlib.h) exported by libaec_amd/lib/libsz.so.2;
same names and argument meaning as reference src/szlib.h:26-43."""
import ctypes as C
import os

import numpy as np

from . import api

SZ_ALLOW_K13_OPTION_MASK = 1
SZ_CHIP_OPTION_MASK = 2
SZ_EC_OPTION_MASK = 4
SZ_LSB_OPTION_MASK = 8
SZ_MSB_OPTION_MASK = 16
SZ_NN_OPTION_MASK = 32
SZ_RAW_OPTION_MASK = 128
SZ_OK = 0
SZ_OUTBUFF_FULL = 2


class SZ_com_t(C.Structure):
    """reference src/szlib.h:26-32"""
    _fields_ = [("options_mask", C.c_int), ("bits_per_pixel", C.c_int),
                ("pixels_per_block", C.c_int), ("pixels_per_scanline", C.c_int)]


def bind(lib):
    for name in ("SZ_BufftoBuffCompress", "SZ_BufftoBuffDecompress"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(SZ_com_t)]
    lib.SZ_encoder_enabled.restype = C.c_int
    return lib


_lib = None


def library_path():
    return os.path.join(os.path.dirname(api.library_path()), "libsz.so.2")


def library():
    global _lib
    if _lib is None:
        api.library()                       # libaec.so.0 (and the process' HIP runtime) first
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run make -C libaec_amd/csrc")
        _lib = bind(C.CDLL(path))
    return _lib


def _call(lib, name, data, out_size, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    out = np.zeros(max(int(out_size), 1), dtype=np.uint8)
    n = C.c_size_t(int(out_size))
    p = SZ_com_t(options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline)
    rc = getattr(lib, name)(out.ctypes.data, C.byref(n), a.ctypes.data, a.size, C.byref(p))
    return rc, out[:n.value].tobytes()


def compress(data, out_size, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline, lib=None):
    """SZ_BufftoBuffCompress (reference src/sz_compat.c:110-183) -> (rc, bytes)"""
    return _call(lib or library(), "SZ_BufftoBuffCompress", data, out_size, options_mask, bits_per_pixel,
                 pixels_per_block, pixels_per_scanline)


def decompress(data, out_size, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline, lib=None):
    """SZ_BufftoBuffDecompress (reference src/sz_compat.c:185-268) -> (rc, bytes)"""
    return _call(lib or library(), "SZ_BufftoBuffDecompress", data, out_size, options_mask, bits_per_pixel,
                 pixels_per_block, pixels_per_scanline)


def _batch(name, chunks, out_sizes, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline):
    """SZ_BatchCompress / SZ_BatchDecompress (extension, include/szlib.h): n chunks in one call.
    Returns (rc, [bytes per chunk], [status per chunk])."""
    lib = library()
    fn = getattr(lib, name)
    fn.restype = C.c_int
    n = len(chunks)
    arrs = [np.frombuffer(bytes(c), dtype=np.uint8) if not isinstance(c, np.ndarray)
            else np.ascontiguousarray(c).view(np.uint8).reshape(-1) for c in chunks]
    outs = [np.zeros(max(int(s), 1), dtype=np.uint8) for s in out_sizes]
    src = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    src_len = (C.c_size_t * n)(*[a.size for a in arrs])
    dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    dst_len = (C.c_size_t * n)(*[int(s) for s in out_sizes])
    status = (C.c_int * n)()
    p = SZ_com_t(options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline)
    rc = fn(dst, dst_len, src, src_len, C.c_size_t(n), C.byref(p), status)
    return rc, [outs[i][:dst_len[i]].tobytes() for i in range(n)], list(status)


def compress_batch(chunks, out_sizes, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline):
    return _batch("SZ_BatchCompress", chunks, out_sizes, options_mask, bits_per_pixel, pixels_per_block,
                  pixels_per_scanline)


def decompress_batch(chunks, out_sizes, options_mask, bits_per_pixel, pixels_per_block, pixels_per_scanline):
    return _batch("SZ_BatchDecompress", chunks, out_sizes, options_mask, bits_per_pixel, pixels_per_block,
                  pixels_per_scanline)
