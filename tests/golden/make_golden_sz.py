#!/usr/bin/env python3
"""Generate tests/golden/sz_vectors.npz with the REFERENCE SZIP shim (src/sz_compat.c inside
oracle/_ref/libaec_ref.so).  Run in the build container only.  Data only: parameters, input
bytes, the bytes SZ_BufftoBuffCompress produced and the length SZ_BufftoBuffDecompress returned."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from helpers import REF_SO  # noqa: E402
from libaec_amd import szip  # noqa: E402

NN, MSB, RAW, LSB = szip.SZ_NN_OPTION_MASK, szip.SZ_MSB_OPTION_MASK, szip.SZ_RAW_OPTION_MASK, szip.SZ_LSB_OPTION_MASK


def main():
    ref = szip.bind(C.CDLL(REF_SO))
    rng = np.random.default_rng(7)
    cases = []

    def walk(n, bits, scale):
        hi = (1 << min(bits, 62)) - 1
        x = np.clip(hi // 2 + np.cumsum(np.rint(rng.standard_normal(n) * scale)), 0, hi).astype(np.uint64)
        x[n // 3: n // 3 + n // 10] = x[n // 3]
        return x

    def add(name, data, opts, bpp, ppb, pps):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        rc, comp = szip.compress(data, data.size * 2 + 4096, opts, bpp, ppb, pps, lib=ref)
        assert rc == 0, (name, rc)
        rc, dec = szip.decompress(comp, data.size, opts, bpp, ppb, pps, lib=ref)
        assert rc == 0 and dec == data.tobytes(), (name, rc, len(dec))
        cases.append((name, (opts, bpp, ppb, pps), data, np.frombuffer(comp, np.uint8)))

    # BASELINE config 5: HDF5-style 8-bit chunks, 8 px/block, 1024 px/scanline
    add("c5-8bit-nn-msb", walk(64 * 1024, 8, 1.5).astype(np.uint8), NN | MSB | RAW, 8, 8, 1024)
    add("c5-8bit-nn-lsb", walk(20000, 8, 4).astype(np.uint8), NN | LSB | RAW, 8, 8, 1024)   # incomplete last line
    add("8bit-ec", walk(5000, 8, 30).astype(np.uint8), RAW, 8, 16, 100)                     # no NN: zero padding
    # 16-bit, scan line not a multiple of the block (padded lines), both byte orders
    add("16bit-pad-lsb", walk(3000, 16, 20).astype("<u2"), NN | LSB | RAW, 16, 32, 1000)
    add("16bit-pad-msb", walk(3000, 14, 20).astype(">u2"), NN | MSB | RAW, 14, 10, 250)
    add("16bit-exact", walk(4096, 12, 5).astype("<u2"), NN | RAW, 12, 16, 256)
    # 32 / 64 bit pixels are coded as byte planes (reference tests/check_szcomp.c: 64 bpp, 8 ppb, 1024 pps)
    add("32bit-planes", walk(25000, 32, 3000).astype("<u4"), NN | RAW, 32, 16, 1000)
    add("64bit-planes", walk(8192, 60, 1e9).astype(">u8"), NN | MSB | RAW, 64, 8, 1024)
    add("24bit", walk(6000, 24, 300).astype("<u4"), NN | RAW, 24, 32, 500)

    names = np.array([c[0] for c in cases])
    params = np.array([c[1] for c in cases], dtype=np.int32)
    in_off = np.cumsum([0] + [c[2].size for c in cases]).astype(np.uint64)
    out_off = np.cumsum([0] + [c[3].size for c in cases]).astype(np.uint64)
    np.savez_compressed(os.path.join(HERE, "sz_vectors.npz"), names=names, params=params, in_off=in_off,
                        out_off=out_off, inputs=np.concatenate([c[2] for c in cases]),
                        outputs=np.concatenate([c[3] for c in cases]))
    print(len(cases), "SZ vectors,", int(in_off[-1]), "input bytes")


if __name__ == "__main__":
    main()
