"""libaec_amd -- MI355X-native CCSDS 121.0-B-2 adaptive entropy coder behind the libaec C ABI.

The product is the shared library ``libaec_amd/lib/libaec.so.0`` (HIP kernels for gfx950 plus the
host stream layer).  This package is only the Python-side mirror of its two interfaces:

* :mod:`libaec_amd.api`  -- ``struct aec_stream`` and the eight libaec entry points
  (reference src/libaec.h:67-166), same names, argument meaning and return codes;
* :mod:`libaec_amd.gpu`  -- the device-resident batch interface (include/aec_gpu.h) for
  callers that keep their buffers in HBM (torch tensors are used for device memory only).

There is no CPU implementation in this package: loading fails loudly when the library has not
been built, and every codec call fails when no HIP device is usable.
"""
from .api import (AEC_CONF_ERROR, AEC_DATA_3BYTE, AEC_DATA_ERROR, AEC_DATA_MSB,  # noqa: F401
                  AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, AEC_FLUSH, AEC_MEM_ERROR, AEC_NO_FLUSH,
                  AEC_NOT_ENFORCE, AEC_OK, AEC_PAD_RSI, AEC_RESTRICTED, AEC_STREAM_ERROR,
                  AecStream, aec_buffer_decode, aec_buffer_encode, library, library_path)

__all__ = [n for n in dir() if n.startswith(("AEC_", "aec_", "Aec", "library"))]
