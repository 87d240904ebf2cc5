"""CPU check of the per-lane device functions (libaec_amd/csrc/aec_lane.h) and of the parallel
reformulation (plateau clamp for k, segment-local zero runs, word assembly) through the host
harness tests/emul/emul.cpp, against the oracle and the reference-produced golden vectors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import (AEC_DATA_3BYTE, AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED,
                     AEC_NOT_ENFORCE, AEC_RESTRICTED, OPT_SE, OPT_SPLIT, OPT_UNCOMP, OPT_ZERO,
                     OPT_ZERO_CONT, ROOT, bytes_per_sample, max_encoded_size, oracle_decode,
                     oracle_encode, pack_samples, random_walk_samples)

EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libemul.so")


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    srcs = [os.path.join(EMUL_DIR, "emul.cpp"),
            os.path.join(ROOT, "libaec_amd", "csrc", "aec_lane.h"),
            os.path.join(ROOT, "libaec_amd", "csrc", "aec_cfg.h"),
            os.path.join(ROOT, "libaec_amd", "csrc", "aec_spec.h"),
            os.path.join(ROOT, "libaec_amd", "csrc", "aec_spec2.h")]
    if not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                        "-o", EMUL_SO, srcs[0]], check=True)
    lib = C.CDLL(EMUL_SO)
    lib.emul_encode.restype = C.c_int
    lib.emul_decode.restype = C.c_int
    lib.emul_index.restype = C.c_int
    return lib


def emul_encode(lib, data, bps, bs, rsi, flags, start_bit=0, k_in=0):
    a = np.ascontiguousarray(data, dtype=np.uint8)
    cap = max_encoded_size(a.size, bps, bs, flags) + 16
    out = np.zeros(cap, np.uint8)
    nb = bytes_per_sample(bps, flags)
    nblk = (a.size // nb + bs - 1) // bs
    nrsi = (nblk + rsi - 1) // rsi
    meta = np.zeros(max(nblk, 1), np.uint32)
    offs = np.zeros(nrsi + 1, np.uint64)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    tb, ko = C.c_uint64(0), C.c_uint32(0)
    rc = lib.emul_encode(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.size), C.c_void_p(out.ctypes.data),
                         C.c_size_t(cap), C.c_uint32(start_bit), C.c_uint32(k_in), C.byref(tb),
                         C.byref(ko), C.c_void_p(meta.ctypes.data), C.c_void_p(offs.ctypes.data))
    return rc, out, tb.value, ko.value, meta[:nblk], offs


def stream_bytes(out, total_bits, start_bit=0):
    n = max(1, (start_bit + total_bits + 7) // 8) if start_bit + total_bits else 1
    return out[:n].tobytes()


def check_case(lib, name, bps, bs, rsi, flags, data, expect):
    rc, out, tb, ko, meta, offs = emul_encode(lib, data, bps, bs, rsi, flags)
    assert rc == 0, name
    assert stream_bytes(out, tb) == expect, name
    # per-block summary must agree with the sequential oracle
    rc, enc, trace, o_offs, o_bits = oracle_encode(data, bps, bs, rsi, flags, want_trace=True)
    assert o_bits == tb, name
    assert np.array_equal(offs[:-1], o_offs), name
    assert np.array_equal(meta & 0xFFF, trace["bits"]), name
    opt_map = {0: OPT_ZERO, 1: OPT_SE, 2: OPT_SPLIT, 3: OPT_UNCOMP, 4: OPT_ZERO_CONT}
    assert [opt_map[int(o)] for o in (meta >> 12) & 7] == trace["option"].tolist(), name
    if len(trace):
        assert ko == trace["k"][-1], name
    # decode from the RSI offsets
    nb = bytes_per_sample(bps, flags)
    nblk = (data.size // nb + bs - 1) // bs
    dec = np.zeros(nblk * bs * nb + 8, np.uint8)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    enc_a = np.frombuffer(expect, dtype=np.uint8)
    rc = lib.emul_decode(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size),
                         C.c_void_p(offs.ctypes.data), C.c_uint64(len(offs) - 1), C.c_uint64(nblk),
                         C.c_void_p(dec.ctypes.data), C.c_size_t(dec.size))
    assert rc == 0, name
    rc, odec, _ = oracle_decode(expect, bps, bs, rsi, flags, nblk * bs * nb)
    assert dec[:nblk * bs * nb].tobytes() == odec, name
    # serial index pass finds the same RSI offsets
    ioffs = np.zeros(len(offs) + 2, np.uint64)
    n_rsi, tail, endb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib.emul_index(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint64(0),
                        C.c_void_p(ioffs.ctypes.data), C.c_uint64(len(ioffs)), C.byref(n_rsi),
                        C.byref(tail), C.byref(endb))
    assert rc == 0, name
    # a rest-of-segment zero run that closes a short final RSI is indistinguishable from one
    # reaching the nominal segment end, so the walker may count up to 63 phantom blocks there
    assert nblk <= n_rsi.value * rsi + tail.value <= nblk + 63, (name, n_rsi.value, tail.value, nblk)
    assert endb.value == tb, name
    nfound = min(n_rsi.value + (1 if tail.value else 0), len(offs) - 1)
    assert np.array_equal(ioffs[:nfound], offs[:nfound]), name
    check_spec(lib, name, bps, bs, rsi, flags, enc_a, offs, nblk)


def check_spec(lib, name, bps, bs, rsi, flags, enc_a, offs, nblk, core=128, look=4096):
    """Speculative index tables (aec_spec.h): at every true RSI start whose RSI fits in the window
    the tabulated RSI length must be the true one; the chained window hops must land on true RSI
    starts.  (Entries at other bit positions are hypotheses nobody reads.)"""
    nbits = enc_a.size * 8
    T = np.zeros(nbits + 1, np.uint16)
    Xb = np.zeros(nbits + 1, np.uint16)
    Xc = np.zeros(nbits + 1, np.uint8)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    rc = lib.emul_spec(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint32(core),
                       C.c_uint32(look), C.c_void_p(T.ctypes.data), C.c_void_p(Xb.ctypes.data),
                       C.c_void_p(Xc.ctypes.data))
    assert rc == 0, name
    full = nblk // rsi                        # RSIs with all their blocks
    o = [int(x) for x in offs]
    for i in range(full):
        start, true_len = o[i], o[i + 1] - o[i]
        if (start % core) + true_len <= core + look:
            assert T[start] == true_len, (name, i, start, int(T[start]), true_len)
        else:
            assert T[start] in (0, true_len), (name, i)
        cnt = int(Xc[start])
        assert i + cnt <= len(o) - 1, (name, i, cnt)      # (a ROS-closed short last RSI may be hopped too)
        assert start + int(Xb[start]) == o[i + cnt], (name, i, cnt)
        if T[start]:
            assert cnt >= 1, (name, i)
    if nblk % rsi:
        # a short last RSI stays unresolved -- unless a rest-of-segment zero run closes it, which no
        # decoder can tell from a run reaching the nominal RSI end (the serial walk counts it too)
        assert T[o[full]] in (0, o[full + 1] - o[full]), name


def test_golden_through_lane_functions(emul, golden):
    for i in range(len(golden)):
        name, bps, bs, rsi, flags, data, expect, _ = golden.case(i)
        if bps == 1 and flags & AEC_DATA_SIGNED:
            continue
        check_case(emul, name, bps, bs, rsi, flags, data, expect)


def test_random_sweep_vs_oracle(emul):
    rng = np.random.default_rng(4242)
    for it in range(600):
        bps = int(rng.integers(1, 33))
        flags = 0
        if rng.random() < 0.75:
            flags |= AEC_DATA_PREPROCESS
        if rng.random() < 0.5:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.4 and bps > 1:
            flags |= AEC_DATA_SIGNED
        if rng.random() < 0.3:
            flags |= AEC_DATA_3BYTE
        if bps <= 4 and rng.random() < 0.5:
            flags |= AEC_RESTRICTED
        if rng.random() < 0.2:
            flags |= AEC_NOT_ENFORCE
            bs = int(rng.integers(1, 33)) * 2
        else:
            bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 128, 130, 300]))
        n = int(rng.integers(1, 5000))
        mode = rng.integers(0, 3)
        if mode == 0:
            vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 1, 5, 50, 1000])),
                                       zero_frac=float(rng.choice([0.05, 0.6])))
        elif mode == 1:
            lo = -(1 << (bps - 1)) if flags & AEC_DATA_SIGNED else 0
            hi = (1 << (bps - 1)) - 1 if flags & AEC_DATA_SIGNED else (1 << bps) - 1
            vals = rng.integers(lo, hi + 1, size=n)
        else:   # long constant stretches: zero runs that close at segment / RSI ends (ROS)
            vals = np.repeat(rng.integers(0, 1 << min(bps, 7), size=n // 97 + 1), 97)[:n]
        data = pack_samples(vals, bps, flags)
        rc, expect, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == 0
        check_case(emul, f"it{it}-n{bps}-j{bs}-r{rsi}-f{flags}-len{n}", bps, bs, rsi, flags, data, expect)


def test_second_extension_table(emul):
    """se_lookup is a closed form; the reference builds the same mapping as a table
    (reference src/decode.c:679-692: for i in 0..12, for k in 0..i: table[ms] = (i, k))."""
    emul.emul_se_lookup.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    ms = 0
    for i in range(13):
        for k in range(i + 1):
            s, t = C.c_uint32(0), C.c_uint32(0)
            assert emul.emul_se_lookup(ms, C.byref(s), C.byref(t)) == 1
            assert (s.value, t.value) == (i, k), ms
            ms += 1
    assert ms == 91
    for m in (91, 92, 1000, 0xFFFFFFFF):
        s, t = C.c_uint32(0), C.c_uint32(0)
        assert emul.emul_se_lookup(m, C.byref(s), C.byref(t)) == 0


def test_carry_in_matches_split_stream(emul):
    """Encoding a stream in two batches with (bit offset, k) carried over must give the bytes
    of the one-shot encode -- what the streaming front-end and the multi-GPU split rely on."""
    rng = np.random.default_rng(9)
    bps, bs, rsi, flags = 16, 16, 8, AEC_DATA_PREPROCESS
    vals = random_walk_samples(rng, bs * rsi * 6, bps, flags, scale=2.0, zero_frac=0.3)
    data = pack_samples(vals, bps, flags)
    rc, whole, *_ = oracle_encode(data, bps, bs, rsi, flags)
    cut = bs * rsi * 2 * 2     # 2 RSIs, in bytes
    rc, o1, tb1, k1, *_ = emul_encode(emul, data[:cut], bps, bs, rsi, flags)
    rc, o2, tb2, k2, *_ = emul_encode(emul, data[cut:], bps, bs, rsi, flags, start_bit=tb1 % 8, k_in=k1)
    nb1 = tb1 // 8
    merged = bytearray(o1[:nb1].tobytes())
    tail = bytearray(stream_bytes(o2, tb2, tb1 % 8))
    if tb1 % 8:
        tail[0] |= int(o1[nb1])
    merged += tail
    assert bytes(merged) == whole


S2_REC = np.dtype([("a", "<u4"), ("x", "<u4"), ("m", "<u4"), ("mx", "<u4")])


@pytest.mark.parametrize("bps,bs,rsi,flags,scale,core,look", [
    (16, 16, 128, AEC_DATA_PREPROCESS, 1.5, 24576, 16384),       # BASELINE config 2 shape
    (8, 8, 128, AEC_DATA_PREPROCESS, 1.5, 24576, 8192),          # config 5 shape
    (12, 16, 40, AEC_DATA_PREPROCESS | AEC_DATA_MSB, 2.0, 16384, 8192),
])
def test_sparse_speculation_vs_oracle(emul, bps, bs, rsi, flags, scale, core, look):
    """Sparse speculative index (aec_spec2.h, what k_spec2 runs per window): on low-entropy streams the
    sync chains must mark (nearly) every true RSI start, and every table value at a true RSI start --
    the RSI length, the chained hop out of the window -- must be exact.  Values at other candidates are
    hypotheses nobody reads."""
    rng = np.random.default_rng(bps + rsi)
    n = bs * rsi * 150 + 7
    vals = random_walk_samples(rng, n, bps, flags, scale=scale, zero_frac=0.1, jump_frac=0.0005)
    data = pack_samples(vals, bps, flags)
    rc, enc, _, offs, bits = oracle_encode(data, bps, bs, rsi, flags)
    enc_a = np.frombuffer(enc, dtype=np.uint8)
    nbits = enc_a.size * 8
    marked = np.zeros(nbits + 64, np.uint8)
    recs = np.zeros(nbits + 64, S2_REC)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    prm = (C.c_uint32 * 8)(core, 4096, look, 64, 8, 0, 0, 0)
    emul.emul_spec2.restype = C.c_int
    rc = emul.emul_spec2(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), prm, C.c_uint64(0),
                         C.c_void_p(marked.ctypes.data), C.c_void_p(recs.ctypes.data))
    assert rc == 0
    o = [int(v) for v in offs] + [bits]
    full = ((n + bs - 1) // bs) // rsi
    miss = 0
    for i in range(full):
        s, true_len = o[i], o[i + 1] - o[i]
        if not marked[s]:
            miss += 1
            continue
        a = int(recs["a"][s])
        assert a in (0, true_len), (i, s, a, true_len)
        if 2 * true_len < look:
            assert a == true_len, (i, s, a, true_len)
        x = int(recs["x"][s])
        cnt, d = x >> 24, x & 0xFFFFFF
        if cnt:
            assert i + cnt <= len(o) - 1 and s + d == o[i + cnt], (i, cnt, d)
    assert marked[0], "the known start of the stream is always a candidate"
    assert miss <= full // 20, (miss, full)       # (a missed start only costs speed: that RSI is walked serially)


@pytest.mark.parametrize("bps,bs,rsi,flags,scale", [
    (16, 16, 128, AEC_DATA_PREPROCESS, 1.5),                     # BASELINE config 2 shape
    (8, 8, 128, AEC_DATA_PREPROCESS, 1.5),                       # config 5 shape
    (32, 32, 64, AEC_DATA_PREPROCESS | AEC_DATA_MSB | AEC_DATA_SIGNED, 40.0),
    (12, 64, 16, AEC_DATA_PREPROCESS, 300.0),                    # long unary parts: many unresolved
    (2, 8, 16, 0, 1.0),                                          # id_len 1
    (24, 16, 7, AEC_DATA_PREPROCESS | AEC_DATA_3BYTE, 3.0),
])
def test_fast_cds_parse_agrees_with_the_full_one(emul, bps, bs, rsi, flags, scale):
    """spec_cds_fast (the straight-line parse the window kernels take first) against spec_cds at every bit
    position, with and without a reference sample: where it answers, it answers the same."""
    rng = np.random.default_rng(bps * 7 + bs)
    n = bs * rsi * 12 + 5
    vals = random_walk_samples(rng, n, bps, flags, scale=scale, zero_frac=0.15, jump_frac=0.002)
    data = pack_samples(vals, bps, flags)
    rc, enc, *_ = oracle_encode(data, bps, bs, rsi, flags)
    enc_a = np.frombuffer(enc, dtype=np.uint8)[: 40000]
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    out = (C.c_uint64 * 4)()
    emul.emul_cds_fast_check.restype = C.c_int
    rc = emul.emul_cds_fast_check(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint32(16384), out)
    assert rc == 0
    assert out[0] > 0 and out[2] == 0, list(out)
    print(f"bps {bps} bs {bs}: {out[0]} positions, {out[1] / out[0]:.3f} unresolved by the fast parse")
