"""ctypes mirror of the libaec C ABI exported by libaec_amd/lib/libaec.so.0.

Names, argument meaning and return codes are those of reference src/libaec.h:67-166; the
helpers at the bottom (`aec_buffer_encode` / `aec_buffer_decode` on Python buffers and the
`Encoder` / `Decoder` streaming wrappers) only fill in a ``struct aec_stream`` the way a C caller
would (e.g. reference src/sz_compat.c:114-175, src/aec.c:149-232).
"""
import ctypes as C
import os

import numpy as np

# reference src/libaec.h:105-124
AEC_DATA_SIGNED = 1
AEC_DATA_3BYTE = 2
AEC_DATA_MSB = 4
AEC_DATA_PREPROCESS = 8
AEC_RESTRICTED = 16
AEC_PAD_RSI = 32
AEC_NOT_ENFORCE = 64
# reference src/libaec.h:129-133
AEC_OK = 0
AEC_CONF_ERROR = -1
AEC_STREAM_ERROR = -2
AEC_DATA_ERROR = -3
AEC_MEM_ERROR = -4
# reference src/libaec.h:141-149
AEC_NO_FLUSH = 0
AEC_FLUSH = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    # AEC_AMD_LIB: another build of the same library (A/B runs of kernel variants, tests/ab_build.sh)
    return os.environ.get("AEC_AMD_LIB") or os.path.join(_HERE, "lib", "libaec.so.0")


class AecStream(C.Structure):
    """struct aec_stream (reference src/libaec.h:67-97)."""
    _fields_ = [
        ("next_in", C.c_void_p),
        ("avail_in", C.c_size_t),
        ("total_in", C.c_size_t),
        ("next_out", C.c_void_p),
        ("avail_out", C.c_size_t),
        ("total_out", C.c_size_t),
        ("bits_per_sample", C.c_uint),
        ("block_size", C.c_uint),
        ("rsi", C.c_uint),
        ("flags", C.c_uint),
        ("state", C.c_void_p),
    ]


_lib = None


def _preload_process_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64.so (same
    SONAME libamdhip64.so.7 as /opt/rocm's); if libaec.so.0 were loaded first it would bind the
    system copy and a later `import torch` would bring in a second runtime, after which HIP calls
    of one of them fail.  Loading torch's copy first makes both resolve to the same object."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def library():
    """Load the HIP-backed libaec.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = library_path()
        _preload_process_hip_runtime()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C libaec_amd/csrc).  libaec_amd has no CPU implementation.")
        lib = C.CDLL(path)
        for name in ("aec_encode_init", "aec_encode_end", "aec_decode_init", "aec_decode_end",
                     "aec_buffer_encode", "aec_buffer_decode"):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(AecStream)]
        for name in ("aec_encode", "aec_decode"):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(AecStream), C.c_int]
        _lib = lib
    return _lib


def bytes_per_sample(bits_per_sample, flags):
    """Container size rule of reference src/encode.c:804-859."""
    if bits_per_sample > 16:
        return 3 if (bits_per_sample <= 24 and flags & AEC_DATA_3BYTE) else 4
    return 2 if bits_per_sample > 8 else 1


def id_len(bits_per_sample, flags):
    if bits_per_sample > 16:
        return 5
    if bits_per_sample > 8:
        return 4
    if flags & AEC_RESTRICTED:
        return 1 if bits_per_sample <= 2 else 2
    return 3


def max_encoded_size(nbytes, bits_per_sample, block_size, flags):
    nsamp = nbytes // bytes_per_sample(bits_per_sample, flags)
    nblk = (nsamp + block_size - 1) // block_size
    return (nblk * (id_len(bits_per_sample, flags) + block_size * bits_per_sample + 2) + 7) // 8 + 64


def _u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _one_shot(fn_name, data, bits_per_sample, block_size, rsi, flags, out_size):
    a = _u8(data)
    out = np.zeros(max(int(out_size), 1), dtype=np.uint8)
    s = AecStream()
    s.next_in, s.avail_in = a.ctypes.data, a.size
    s.next_out, s.avail_out = out.ctypes.data, int(out_size)
    s.bits_per_sample, s.block_size, s.rsi, s.flags = bits_per_sample, block_size, rsi, flags
    rc = getattr(library(), fn_name)(C.byref(s))
    return rc, out[:s.total_out].tobytes() if rc in (AEC_OK, AEC_STREAM_ERROR) else b"", s


def aec_buffer_encode(data, bits_per_sample, block_size, rsi, flags, out_size=None):
    """reference src/encode.c:950-963.  Returns (return code, encoded bytes)."""
    a = _u8(data)
    if out_size is None:
        out_size = max_encoded_size(a.size, bits_per_sample, max(block_size, 1), flags)
    rc, out, _ = _one_shot("aec_buffer_encode", a, bits_per_sample, block_size, rsi, flags, out_size)
    return rc, out


def aec_buffer_decode(data, bits_per_sample, block_size, rsi, flags, out_size):
    """reference src/decode.c:843-854.  Returns (return code, decoded bytes)."""
    rc, out, _ = _one_shot("aec_buffer_decode", data, bits_per_sample, block_size, rsi, flags, out_size)
    return rc, out


class _Stream:
    """Streaming use of the ABI: init / repeated calls with caller-sized chunks / end."""
    _init = _call = _end = None

    def __init__(self, bits_per_sample, block_size, rsi, flags):
        self.lib = library()
        self.s = AecStream()
        self.s.bits_per_sample, self.s.block_size, self.s.rsi, self.s.flags = (
            bits_per_sample, block_size, rsi, flags)
        rc = getattr(self.lib, self._init)(C.byref(self.s))
        if rc != AEC_OK:
            raise ValueError(f"{self._init} failed with {rc}")
        self.open = True

    def call(self, data, out_room, flush=AEC_NO_FLUSH):
        """One aec_encode / aec_decode call.  Returns (rc, bytes consumed, output bytes)."""
        a = _u8(data)
        out = np.zeros(max(out_room, 1), dtype=np.uint8)
        self.s.next_in, self.s.avail_in = a.ctypes.data, a.size
        self.s.next_out, self.s.avail_out = out.ctypes.data, out_room
        rc = getattr(self.lib, self._call)(C.byref(self.s), flush)
        return rc, a.size - self.s.avail_in, out[:out_room - self.s.avail_out].tobytes()

    def end(self):
        if self.open:
            self.open = False
            return getattr(self.lib, self._end)(C.byref(self.s))
        return AEC_OK

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


class Encoder(_Stream):
    _init, _call, _end = "aec_encode_init", "aec_encode", "aec_encode_end"


class Decoder(_Stream):
    _init, _call, _end = "aec_decode_init", "aec_decode", "aec_decode_end"
