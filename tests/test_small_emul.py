"""The every-bit index scheme (libaec_amd/csrc/aec_small.h) on the CPU: tests/emul/small_emul.cpp runs the per-bit
functions the kernels of aec_idx.hip (launch_index_small) are loops over -- the parse at every bit, the hops, the walk
of one RSI from every bit, the doubling in base 4 -- and the chain of RSI starts must be the one the oracle's encoder
reports (reference src/decode.c:402-421, 518-544 is what the walk restates); the walk through the hop table must be the
walk without it at EVERY bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libsmall_emul.so")
PP, MSB, SGN = helpers.AEC_DATA_PREPROCESS, helpers.AEC_DATA_MSB, helpers.AEC_DATA_SIGNED


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    srcs = [os.path.join(EMUL_DIR, "small_emul.cpp")] + [os.path.join(ROOT, "libaec_amd", "csrc", h) for h in
                                                          ("aec_small.h", "aec_trunk.h", "aec_spec.h", "aec_lane.h", "aec_cfg.h")]
    if not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"),
                        "-o", EMUL_SO, srcs[0]], check=True)
    lib = C.CDLL(EMUL_SO)
    lib.emul_small.restype = C.c_int
    return lib


@pytest.mark.parametrize("bps,bs,rsi,flags,n,scale,zero_frac", [
    (8, 8, 1, PP, 3000, 3.0, 0.0),
    (8, 8, 4, PP, 6000, 1.0, 0.3),            # narrow scan lines
    (8, 16, 1, 0, 5000, 3.0, 0.1),            # no preprocessor
    (16, 16, 16, PP | MSB, 12001, 60.0, 0.3),  # a short last RSI
    (16, 8, 64, PP, 9000, 0.3, 0.7),           # long runs of zero blocks: rest-of-segment codes end the hops
    (32, 16, 5, PP | SGN, 4000, 400.0, 0.1),
    (16, 16, 128, 0, 9000, 8.0, 0.5),          # no preprocessor, two segments per RSI
    (24, 32, 33, PP | MSB, 7000, 8.0, 0.3),
    (16, 8, 1024, PP, 30000, 3.0, 0.5),        # RSIs of hundreds of blocks: hops of hops
    (8, 8, 4096, 0, 70000, 1.0, 0.3),
])
def test_every_bit_scheme_finds_the_encoders_rsi_starts(emul, bps, bs, rsi, flags, n, scale, zero_frac):
    rng = np.random.default_rng(bps * 1000 + rsi)
    vals = helpers.random_walk_samples(rng, n, bps, flags, scale=scale, zero_frac=zero_frac)
    raw = np.frombuffer(helpers.pack_samples(vals, bps, flags), dtype=np.uint8)
    rc, enc, _, offs, total_bits = helpers.oracle_encode(raw, bps, bs, rsi, flags)
    assert rc == helpers.AEC_OK
    nblk = (n + bs - 1) // bs
    whole = nblk // rsi                        # (a short last RSI is not a start the chain reaches the END of)
    want = np.asarray(offs, dtype=np.uint64)
    enc_a = np.frombuffer(enc, dtype=np.uint8)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    for hops in (0, 1, 2):
        stats = np.zeros(4, dtype=np.uint64)
        rc = emul.emul_small(p, C.c_void_p(enc_a.ctypes.data), C.c_size_t(enc_a.size), C.c_void_p(want.ctypes.data),
                             C.c_uint64(len(want)), C.c_uint32(hops), C.c_void_p(stats.ctypes.data))
        m, bad, differ, levels = (int(x) for x in stats)
        assert rc == 0 and bad == 0 and differ == 0, (hops, m, bad, differ)
        # every start of a whole RSI is on the chain (the chain may run on into the padding of the last byte)
        assert m >= min(len(want), whole + (1 if nblk % rsi else 0)), (hops, m, len(want), whole)
