"""Entries by plausibility (aec_idx.hip: k_lock_guess_p) modelled on the CPU: tests/emul/plaus_emul.cpp
guesses the entry of a region -- first boundary at or behind its start and the count of blocks there -- from the options
along chains of coded data sets, as the kernel does, on the reference's sample file (tests/golden/typical.rz: 16-bit,
blocks of 64, rsi 256), four regions per RSI.  Nothing rests on a guess in the product (the phase-locked scheme checks
and repairs every entry, and a judge hands streams whose options say nothing to the other schemes); what this pins is
that on real data nine guesses in ten are right, which is what makes the scheme pay."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_SO = os.path.join(EMUL_DIR, "_build", "libplaus_emul.so")


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    srcs = [os.path.join(EMUL_DIR, "plaus_emul.cpp")] + [os.path.join(ROOT, "libaec_amd", "csrc", h) for h in
                                                          ("aec_trunk.h", "aec_spec.h", "aec_lane.h", "aec_cfg.h")]
    if not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"),
                        "-o", EMUL_SO, srcs[0]], check=True)
    lib = C.CDLL(EMUL_SO)
    lib.emul_plaus.restype = C.c_int
    return lib


def test_guesses_on_the_sample_file(emul):
    enc = np.fromfile(os.path.join(ROOT, "tests", "golden", "typical.rz"), dtype=np.uint8)
    p = (C.c_uint32 * 4)(16, 64, 256, 8 | 4)
    stats = np.zeros(4, dtype=np.uint64)
    rc = emul.emul_plaus(p, C.c_void_p(enc.ctypes.data), C.c_size_t(enc.size), C.c_uint32(4), C.c_void_p(stats.ctypes.data))
    regions, right, none, wrong = (int(x) for x in stats)
    assert rc == 0 and regions == 120, (rc, regions)
    # (round 5: 115 right, 1 without a guess in a noisy stretch, 4 wrong; the kernel abandons the scheme for a stream if
    # more than a quarter of its regions disagree with their neighbours after the first walk)
    assert right >= 108 and wrong <= 6, (right, none, wrong)
