"""Shared test plumbing: ctypes bindings for the oracle (oracle/_build/libaec_oracle.so),
the compiled reference (oracle/_ref/libaec_ref.so, optional) and small data generators.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() use this module;
the product package (libaec_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libaec_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libaec_ref.so")

# flag values: reference src/libaec.h:105-124
AEC_DATA_SIGNED = 1
AEC_DATA_3BYTE = 2
AEC_DATA_MSB = 4
AEC_DATA_PREPROCESS = 8
AEC_RESTRICTED = 16
AEC_PAD_RSI = 32
AEC_NOT_ENFORCE = 64

AEC_OK = 0
AEC_CONF_ERROR = -1
AEC_STREAM_ERROR = -2
AEC_DATA_ERROR = -3
AEC_MEM_ERROR = -4

AEC_NO_FLUSH = 0
AEC_FLUSH = 1

OPT_ZERO, OPT_SE, OPT_SPLIT, OPT_UNCOMP, OPT_ZERO_CONT = range(5)


class AecStream(C.Structure):
    """struct aec_stream, reference src/libaec.h:67-97 (72 bytes on x86-64)."""
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


class OracleParams(C.Structure):
    _fields_ = [("bits_per_sample", C.c_uint), ("block_size", C.c_uint),
                ("rsi", C.c_uint), ("flags", C.c_uint)]


class OracleTrace(C.Structure):
    _fields_ = [("option", C.c_uint8), ("k", C.c_uint8), ("reserved", C.c_uint16),
                ("bits", C.c_uint32)]


TRACE_DTYPE = np.dtype([("option", "u1"), ("k", "u1"), ("reserved", "<u2"), ("bits", "<u4")])


class OracleDerived(C.Structure):
    _fields_ = [("id_len", C.c_int), ("bytes_per_sample", C.c_int), ("kmax", C.c_int),
                ("xmin", C.c_uint32), ("xmax", C.c_uint32)]


def build_oracle():
    """(Re)build the checkers with oracle/Makefile; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.aeco_encode.restype = C.c_int
        lib.aeco_encode.argtypes = [C.POINTER(OracleParams), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        lib.aeco_decode.restype = C.c_int
        lib.aeco_decode.argtypes = [C.POINTER(OracleParams), C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_uint64)]
        lib.aeco_derive.restype = C.c_int
        lib.aeco_derive.argtypes = [C.POINTER(OracleParams), C.c_int, C.POINTER(OracleDerived)]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    """The reference libaec itself (built here from /root/reference; travels as a .so)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            build_oracle()
        lib = C.CDLL(REF_SO)
        for name in ("aec_encode_init", "aec_encode_end", "aec_decode_init", "aec_decode_end",
                     "aec_buffer_encode", "aec_buffer_decode"):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(AecStream)]
        for name in ("aec_encode", "aec_decode"):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.POINTER(AecStream), C.c_int]
        _ref = lib
    return _ref


def derive(bps, bs, rsi, flags, for_encode=True):
    p = OracleParams(bps, bs, rsi, flags)
    d = OracleDerived()
    rc = oracle_lib().aeco_derive(C.byref(p), int(for_encode), C.byref(d))
    return rc, d


def bytes_per_sample(bps, flags):
    if bps > 16:
        return 3 if (bps <= 24 and flags & AEC_DATA_3BYTE) else 4
    return 2 if bps > 8 else 1


def id_len_of(bps, flags):
    if bps > 16:
        return 5
    if bps > 8:
        return 4
    if flags & AEC_RESTRICTED:
        return 1 if bps <= 2 else 2
    return 3


def max_encoded_size(nbytes, bps, bs, flags):
    """Safe output capacity: every block is at most id_len + bs*bps bits (+ slack)."""
    bpsb = bytes_per_sample(bps, flags)
    nsamp = nbytes // bpsb
    nblk = (nsamp + bs - 1) // bs
    return (nblk * (id_len_of(bps, flags) + bs * bps) + 7) // 8 + 64


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return np.ascontiguousarray(a)


def oracle_encode(data, bps, bs, rsi, flags, want_trace=False, out_cap=None):
    """Returns (rc, bytes, trace|None, rsi_bit_offsets, total_bits)."""
    a = _as_u8(data)
    cap = max_encoded_size(a.size, bps, bs, flags) if out_cap is None else out_cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    bpsb = bytes_per_sample(bps, flags)
    nsamp = a.size // bpsb
    nblk = (nsamp + bs - 1) // bs if bs else 0
    nrsi = (nblk + rsi - 1) // rsi if rsi else 0
    trace = np.zeros(max(nblk, 1), dtype=TRACE_DTYPE) if want_trace else None
    offs = np.zeros(max(nrsi, 1), dtype=np.uint64)
    n = C.c_size_t(0)
    tb = C.c_uint64(0)
    p = OracleParams(bps, bs, rsi, flags)
    rc = oracle_lib().aeco_encode(C.byref(p), a.ctypes.data, a.size, out.ctypes.data, cap,
                                  C.byref(n), trace.ctypes.data if want_trace else None,
                                  offs.ctypes.data, C.byref(tb))
    return rc, out[:n.value].tobytes(), (trace[:nblk] if want_trace else None), offs[:nrsi], tb.value


def oracle_decode(data, bps, bs, rsi, flags, out_cap):
    a = _as_u8(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    used = C.c_uint64(0)
    p = OracleParams(bps, bs, rsi, flags)
    rc = oracle_lib().aeco_decode(C.byref(p), a.ctypes.data, a.size, out.ctypes.data, out_cap,
                                  C.byref(n), C.byref(used))
    return rc, out[:n.value].tobytes(), used.value


def _stream_call(lib, fn_name, data, bps, bs, rsi, flags, out_cap):
    a = _as_u8(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    s = AecStream()
    s.next_in = a.ctypes.data
    s.avail_in = a.size
    s.next_out = out.ctypes.data
    s.avail_out = out_cap
    s.bits_per_sample, s.block_size, s.rsi, s.flags = bps, bs, rsi, flags
    rc = getattr(lib, fn_name)(C.byref(s))
    return rc, out[:s.total_out].tobytes() if rc in (AEC_OK, AEC_STREAM_ERROR) else b"", s


def ref_encode(data, bps, bs, rsi, flags, out_cap=None):
    """aec_buffer_encode of the compiled reference.  Returns (rc, bytes)."""
    a = _as_u8(data)
    cap = max_encoded_size(a.size, bps, bs, flags) if out_cap is None else out_cap
    rc, out, _ = _stream_call(ref_lib(), "aec_buffer_encode", a, bps, bs, rsi, flags, cap)
    return rc, out


def ref_decode(data, bps, bs, rsi, flags, out_cap):
    rc, out, _ = _stream_call(ref_lib(), "aec_buffer_decode", data, bps, bs, rsi, flags, out_cap)
    return rc, out


# --------------------------------------------------------------------------------------
# data generators
# --------------------------------------------------------------------------------------

def pack_samples(values, bps, flags):
    """Store integer sample values in the container layout selected by bps/flags
    (reference encode_accessors.c:61-143)."""
    v = np.asarray(values).astype(np.int64) & ((1 << bps) - 1)
    nb = bytes_per_sample(bps, flags)
    out = np.empty((v.size, nb), dtype=np.uint8)
    for i in range(nb):
        shift = 8 * (nb - 1 - i) if flags & AEC_DATA_MSB else 8 * i
        out[:, i] = (v >> shift) & 0xFF
    return out.reshape(-1)


def unpack_samples(buf, bps, flags):
    nb = bytes_per_sample(bps, flags)
    b = _as_u8(buf)
    b = b[: (b.size // nb) * nb].reshape(-1, nb).astype(np.int64)
    v = np.zeros(b.shape[0], dtype=np.int64)
    for i in range(nb):
        shift = 8 * (nb - 1 - i) if flags & AEC_DATA_MSB else 8 * i
        v |= b[:, i] << shift
    return v


def random_walk_samples(rng, n, bps, flags, scale=3.0, zero_frac=0.1, jump_frac=0.01):
    """Mixed-entropy data: random walk with occasional constant runs (zero blocks after
    preprocessing) and occasional full-range jumps (uncompressed/high-k blocks)."""
    signed = bool(flags & AEC_DATA_SIGNED)
    lo = -(1 << (bps - 1)) if signed else 0
    hi = (1 << (bps - 1)) - 1 if signed else (1 << bps) - 1
    steps = np.rint(rng.standard_normal(n) * scale * rng.choice([0.2, 1, 8, 200], size=n,
                    p=[0.4, 0.4, 0.15, 0.05])).astype(np.int64)
    # constant stretches
    i = 0
    while i < n:
        if rng.random() < zero_frac:
            ln = int(rng.integers(1, 400))
            steps[i:i + ln] = 0
            i += ln
        else:
            i += int(rng.integers(1, 200))
    jumps = rng.random(n) < jump_frac
    x = np.empty(n, dtype=np.int64)
    cur = (lo + hi) // 2
    jump_vals = rng.integers(lo, hi + 1, size=n)
    for j in range(n):
        cur = int(jump_vals[j]) if jumps[j] else min(hi, max(lo, cur + int(steps[j])))
        x[j] = cur
    return x


def craft_overlong_stream(rng, bps, bs, rsi, n_rsi, id_len, long_frac, long_hi, pp=True):
    """A VALID CCSDS 121 stream no libaec encoder would write: some blocks take split option k = 0 for large
    mapped residuals, so that their coded data sets are many times longer than the uncompressed option
    (fundamental sequences of up to `long_hi` zeros each).  Returns the bytes."""
    bits = []

    def put(v, n):
        bits.extend(((v >> (n - 1 - i)) & 1) for i in range(n))

    for r in range(n_rsi):
        for b in range(rsi):
            ref = pp and b == 0
            long_one = rng.random() < long_frac
            k = 0 if long_one else int(rng.integers(0, 3))
            put(k + 1, id_len)                                   # split option k
            if ref:
                put(int(rng.integers(0, 1 << bps)), bps)
            vals = rng.integers(0, long_hi if long_one else 6, bs - (1 if ref else 0))
            for v in vals:
                put(1, int(v >> k) + 1)                          # fundamental sequence: zeros, then a one
            if k:
                for v in vals:
                    put(int(v) & ((1 << k) - 1), k)
    while len(bits) % 8:
        bits.append(0)
    return np.packbits(np.array(bits, dtype=np.uint8)).tobytes()
