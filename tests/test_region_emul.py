"""The region index (libaec_amd/csrc/aec_region.h, aec_region.hip) on the CPU: tests/emul/region_emul.cpp
runs the scheme's per-lane functions region by region against the RSI starts the oracle's encoder reports.

* the lane's parsers (register window, ring) return what the exact parse from memory returns, at every bit;
* the guesses are mostly right on the benchmark shapes and on the reference's sample file (a wrong guess only costs a second
  walk of its region in the product -- this pins that the scheme pays);
* the whole pass -- guesses, walks, every entry checked against the walk in front, mending passes, RSI starts -- delivers
  the oracle's table, also when a third of the guesses or every single one is made wrong on purpose."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libregion_emul.so")
PP, MSB, SIGNED = 8, 4, 1


class GenState(C.Structure):
    _fields_ = [("s", C.c_uint64), ("x", C.c_int64), ("i", C.c_uint64), ("hold", C.c_uint32), ("kind", C.c_uint32)]


def generate(kind, nbytes):
    """libaec_amd/csrc/datagen.c: 0 lowent16, 1 lowent32s, 2 chunks8 (the benchmark inputs)."""
    gen = C.CDLL(os.path.join(ROOT, "libaec_amd", "lib", "libaec_datagen.so"))
    st = GenState()
    gen.aec_gen_init(C.byref(st), C.c_uint(kind), C.c_uint64(0))
    out = np.zeros(nbytes, dtype=np.uint8)
    gen.aec_gen_fill(C.byref(st), C.c_void_p(out.ctypes.data), C.c_size_t(nbytes // [2, 4, 1][kind]))
    return out


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    srcs = [os.path.join(EMUL_DIR, "region_emul.cpp")] + [os.path.join(ROOT, "libaec_amd", "csrc", h) for h in
                                                           ("aec_region.h", "aec_trunk.h", "aec_spec.h", "aec_lane.h", "aec_cfg.h")]
    if not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"),
                        "-o", EMUL_SO, srcs[0]], check=True)
    lib = C.CDLL(EMUL_SO)
    lib.emul_region_parser.restype = C.c_uint64
    return lib


def sample_file():
    enc = np.fromfile(os.path.join(ROOT, "tests", "golden", "typical.rz"), dtype=np.uint8)
    rc, dec, _ = H.oracle_decode(enc, 16, 64, 256, PP | MSB, 1 << 20)
    assert rc == 0
    return np.frombuffer(dec, dtype=np.uint8)


SHAPES = {
    # name: (data, bits per sample, block, rsi, flags, region bits, budget, least share of right guesses)
    "config 2": (lambda: generate(0, 8 << 20), 16, 16, 128, PP, 32768, 1152, 0.97),
    "config 5": (lambda: generate(2, 4 << 20), 8, 8, 128, PP, 32768, 1152, 0.85),
    "config 3": (lambda: generate(1, 32 << 20), 32, 32, 4096, PP | MSB | SIGNED, 1 << 20, 8832, 0.9),
    "the sample file": (lambda: np.tile(sample_file(), 4), 16, 64, 256, PP | MSB, 1 << 18, 4000, 0.9),
}


def encode(name):
    make, bps, bs, rsi, flags, region, budget, least = SHAPES[name]
    data = make()
    rc, enc, _, offs, total_bits = H.oracle_encode(data, bps, bs, rsi, flags)
    assert rc == 0
    return np.frombuffer(enc, dtype=np.uint8), np.ascontiguousarray(offs), total_bits


@pytest.mark.parametrize("name", list(SHAPES))
def test_lane_parsers_against_the_parse_from_memory(emul, name):
    _, bps, bs, rsi, flags, _, _, _ = SHAPES[name]
    enc, _, _ = encode(name)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    n = min(enc.size, 60_000)
    assert emul.emul_region_parser(p, C.c_void_p(enc.ctypes.data), C.c_size_t(n), C.c_uint64(1)) == 0
    # (and the last bytes of the stream: reads beyond the buffer, coded data sets that the input cuts)
    tail = np.ascontiguousarray(enc[-4099:])
    assert emul.emul_region_parser(p, C.c_void_p(tail.ctypes.data), C.c_size_t(tail.size), C.c_uint64(1)) == 0


@pytest.mark.parametrize("name", list(SHAPES))
def test_guesses(emul, name):
    _, bps, bs, rsi, flags, region, budget, least = SHAPES[name]
    enc, offs, _ = encode(name)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    stats = np.zeros(16, dtype=np.uint64)
    rc = emul.emul_region_guess(p, C.c_void_p(enc.ctypes.data), C.c_size_t(enc.size), C.c_void_p(offs.ctypes.data),
                                C.c_size_t(offs.size), C.c_uint64(region), C.c_uint32(budget), C.c_void_p(stats.ctypes.data), None)
    regions, right, none, wrong = (int(x) for x in stats[:4])
    assert rc == 0 and regions > 20
    assert right >= least * regions and wrong <= 0.1 * regions, (regions, right, none, wrong)


@pytest.mark.parametrize("name,sabotage_every,shift,passes", [
    ("config 2", 0, 0, 12), ("config 5", 0, 0, 12), ("config 3", 0, 0, 12), ("the sample file", 0, 0, 12),
    ("config 2", 3, 45, 12),            # every third guess 45 bits off
    ("config 5", 3, -7, 24),            # every third (with the 8-bit data's own wrong guesses: runs of them)
    ("config 2", 1, -100, 4000),        # every single one: a region per pass
])
def test_the_whole_pass_delivers_the_oracles_table(emul, name, sabotage_every, shift, passes):
    _, bps, bs, rsi, flags, region, budget, _ = SHAPES[name]
    enc, offs, total_bits = encode(name)
    if sabotage_every == 1:             # (a serial chain of repairs: a short stream)
        nr = 512
        enc, total_bits, offs = enc[: int(offs[nr] + 7) // 8], int(offs[nr]), offs[:nr]
        enc = np.ascontiguousarray(enc)
    p = (C.c_uint32 * 4)(bps, bs, rsi, flags)
    stats = np.zeros(16, dtype=np.uint64)
    out = np.zeros(offs.size + 8, dtype=np.uint64)
    rc = emul.emul_region_index(p, C.c_void_p(enc.ctypes.data), C.c_size_t(enc.size), C.c_uint64(region), C.c_uint32(budget),
                                C.c_uint32(passes), C.c_uint32(sabotage_every), C.c_int64(shift), C.c_void_p(out.ctypes.data),
                                C.c_uint64(out.size), C.c_void_p(stats.ctypes.data))
    nreg, kept, differ, busy, delivered, n_rsi, tail, end = (int(x) for x in stats[:8])
    assert rc == 0 and delivered == 1, (nreg, kept, differ, busy)
    if sabotage_every:
        assert differ > 0 and busy > 0
    assert np.array_equal(out[: offs.size], offs)
