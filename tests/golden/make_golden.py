#!/usr/bin/env python3
"""Generate tests/golden/vectors.npz with the REFERENCE libaec (oracle/_ref/libaec_ref.so,
compiled from /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Every vector is data: the stream parameters, the input bytes and the bytes the reference's
aec_buffer_encode produced for them (plus the length aec_buffer_decode returns).  The
reference's shipped known-answer file data/typical.rz is copied next to it as a data
fixture.  No reference source text is stored.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import (AEC_DATA_3BYTE, AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED,  # noqa: E402
                     AEC_NOT_ENFORCE, AEC_RESTRICTED, bytes_per_sample, pack_samples,
                     random_walk_samples, ref_decode, ref_encode)

PP, MSB, SGN, B3, RES, NE = (AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED,
                             AEC_DATA_3BYTE, AEC_RESTRICTED, AEC_NOT_ENFORCE)


def limits(bps, flags):
    if flags & SGN:
        return -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    return 0, (1 << bps) - 1


def pattern(kind, n, bps, flags, k=0):
    """The crafted inputs of reference tests/check_code_options.c:37-195."""
    xmin, xmax = limits(bps, flags)
    pp = bool(flags & PP)
    if kind == "zero":            # :41-44
        byte = 0x55 if pp else 0
        return np.full(n * bytes_per_sample(bps, flags), byte, dtype=np.uint8)
    if kind == "se":              # :160-186
        cell = [xmax - 1] * 4 + [xmax] * 4 if pp else [0, 0, 0, 0, 1, 0, 0, 2]
    elif kind == "uncomp":        # :99-104
        cell = [xmax, xmin]
    elif kind == "fs":            # :124-142
        cell = [xmin + 2, xmin, xmin, xmin] if pp else [0, 0, 0, 4]
    elif kind == "split":         # :62-81
        cell = ([xmin + (1 << (k - 1)) - 1, xmin, xmin + (1 << (k + 1)) - 1, xmin] if pp
                else [0, (1 << k) - 1, 0, (1 << (k + 2)) - 1])
    else:
        raise ValueError(kind)
    vals = np.tile(np.array(cell, dtype=np.int64), (n + len(cell) - 1) // len(cell))[:n]
    return pack_samples(vals, bps, flags)


def main():
    rng = np.random.default_rng(20261002)
    cases = []

    def add(name, data, bps, bs, rsi, flags):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        rc, enc = ref_encode(data, bps, bs, rsi, flags)
        assert rc == 0, (name, rc)
        nb = bytes_per_sample(bps, flags)
        nblk = (data.size // nb + bs - 1) // bs
        rc, dec = ref_decode(enc, bps, bs, rsi, flags, nblk * bs * nb)
        assert rc == 0, (name, rc)
        cases.append((name, (bps, bs, rsi, flags), data, np.frombuffer(enc, dtype=np.uint8),
                      len(dec)))

    # 1. code-option patterns (check_code_options.c) over the byte-order / sign sweep
    sweeps = [0, PP, PP | SGN, PP | MSB, PP | MSB | SGN]
    for bps in (8, 16, 24, 32):
        for fl in sweeps:
            f = fl | (B3 if bps == 24 else 0)
            for bs, rsi in ((8, 3), (16, 128), (64, 5)):
                n = bs * rsi * 2
                for kind in ("zero", "se", "uncomp", "fs"):
                    add(f"opt-{kind}-n{bps}-j{bs}-r{rsi}-f{f}", pattern(kind, n, bps, f), bps, bs, rsi, f)
                for k in sorted({1, 2, bps // 2, bps - 3}):
                    add(f"opt-split{k}-n{bps}-j{bs}-r{rsi}-f{f}", pattern("split", n, bps, f, k),
                        bps, bs, rsi, f)

    # 2. state->k carry probe: eight 2s after eight 12s, across an RSI boundary
    add("kcarry-first", np.array([2] * 8, np.uint8), 8, 8, 1, 0)
    add("kcarry-after12", np.array([12] * 8 + [2] * 8, np.uint8), 8, 8, 1, 0)
    add("kcarry-after12-r2", np.array([12] * 8 + [2] * 8, np.uint8), 8, 8, 2, 0)

    # 3. tails and degenerate sizes (check_buffer_sizes.c:24-47 and friends)
    add("empty", np.zeros(0, np.uint8), 16, 16, 128, PP)
    add("one-sample", pack_samples([1234], 16, PP), 16, 16, 128, PP)
    add("11-samples", pack_samples(np.arange(11) * 3, 16, PP), 16, 16, 128, PP)
    add("odd-trailing-byte", np.concatenate([pack_samples(np.arange(40), 16, PP), np.zeros(1, np.uint8)]),
        16, 16, 2, PP)
    xmin, xmax = limits(32, PP)
    alt = np.tile(np.array([xmax, xmin], dtype=np.int64), 16 * 4 * 2)
    add("bufsize-full", pack_samples(alt[:16 * 4 * 2], 32, PP), 32, 16, 4, PP)
    add("bufsize-short", pack_samples(alt[:16 * 4 * 2 - 2 * 16 + 1], 32, PP), 32, 16, 4, PP)
    # check_long_fs.c:8-29
    lf = np.array([0] * 32 + [65000] * 32, dtype=np.int64)
    add("long-fs", pack_samples(lf, 16, PP), 16, 64, 1, PP)

    # 4. randomized mixes over the parameter space (incl. non-byte bps, RESTRICTED, NOT_ENFORCE)
    combos = [
        (16, 16, 128, PP), (16, 64, 256, PP | MSB), (32, 32, 4096, PP | MSB | SGN), (8, 8, 128, PP),
        (8, 8, 128, PP | MSB), (12, 16, 64, PP | SGN), (24, 32, 10, PP | B3), (24, 32, 10, PP | B3 | MSB | SGN),
        (20, 8, 65, PP), (32, 64, 130, 0), (1, 8, 16, PP | RES), (2, 16, 9, RES), (3, 8, 70, PP | RES),
        (4, 32, 4, PP | RES | SGN), (5, 8, 64, PP), (10, 10, 20, PP | NE), (16, 2, 300, PP | NE),
        (7, 8, 1, PP | SGN), (17, 16, 200, PP | SGN | MSB), (32, 8, 64, PP | SGN),
    ]
    for bps, bs, rsi, fl in combos:
        for scale, zf, nmul in ((0.4, 0.5, 2.3), (3.0, 0.1, 1.0), (300.0, 0.02, 0.37)):
            n = max(1, int(bs * rsi * nmul))
            n = min(n, 6000)
            vals = random_walk_samples(rng, n, bps, fl, scale=scale, zero_frac=zf)
            add(f"mix-n{bps}-j{bs}-r{rsi}-f{fl}-s{scale}", pack_samples(vals, bps, fl), bps, bs, rsi, fl)

    names = np.array([c[0] for c in cases])
    params = np.array([c[1] for c in cases], dtype=np.uint32)
    declen = np.array([c[4] for c in cases], dtype=np.uint64)
    in_off = np.cumsum([0] + [c[2].size for c in cases]).astype(np.uint64)
    out_off = np.cumsum([0] + [c[3].size for c in cases]).astype(np.uint64)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), names=names, params=params,
                        decoded_len=declen, in_off=in_off, out_off=out_off,
                        inputs=np.concatenate([c[2] for c in cases]),
                        outputs=np.concatenate([c[3] for c in cases]))
    shutil.copyfile("/root/reference/data/typical.rz", os.path.join(HERE, "typical.rz"))
    print(f"{len(cases)} vectors, {int(in_off[-1])} input bytes, {int(out_off[-1])} output bytes")


if __name__ == "__main__":
    main()
