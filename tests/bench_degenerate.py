#!/usr/bin/env python3
"""Degenerate inputs through aec_buffer_encode / aec_buffer_decode (run on the GPU box): all zeros, a constant, a ramp,
incompressible noise, and noise in the low bits of a constant -- the streams whose coded data sets are as short or as
long as the format allows.  Product against the compiled reference on one core.

    python tests/bench_degenerate.py [--size-mib 64]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-mib", type=int, default=64)
    ap.add_argument("--only", default="", help="one shape only (zeros, constant, ramp, noise, ...)")
    ap.add_argument("--bps", type=int, default=0, help="16 or 8 only")
    args = ap.parse_args()
    import torch  # noqa: F401
    from helpers import have_ref, ref_decode
    from libaec_amd import api
    n = args.size_mib << 20
    rng = np.random.default_rng(5)
    for bps, bs, rsi in ((16, 16, 128), (8, 8, 128)):
        if args.bps and args.bps != bps:
            continue
        dt = np.dtype("<u2") if bps == 16 else np.dtype(np.uint8)
        m = n // dt.itemsize
        shapes = {
            "zeros": np.zeros(m, dtype=dt),
            "constant": np.full(m, 1000 if bps == 16 else 100, dtype=dt),
            "ramp": (np.arange(m) % (1 << bps)).astype(dt),
            "noise": rng.integers(0, 1 << bps, m, dtype=np.uint64).astype(dt),
            "noise in 2 low bits": (rng.integers(0, 4, m, dtype=np.uint64) + (1 << (bps - 1))).astype(dt),
        }
        for name, arr in shapes.items():
            if args.only and name != args.only:
                continue
            data = arr.view(np.uint8)
            flags = api.AEC_DATA_PREPROCESS
            t0 = time.perf_counter()
            rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
            t1 = time.perf_counter()
            assert rc == 0
            # The call itself, into a caller's buffer that exists (pages touched), as a C caller's does: the Python wrapper's
            # fresh np.zeros + .tobytes() add two passes of first-touch page faults over 64 MiB (the "20 ms floor" of the
            # round-5 table was that, not the library).
            a = np.frombuffer(enc, dtype=np.uint8)
            out = np.ones(n, dtype=np.uint8)
            best = 1e9
            for _ in range(3):
                s = api.AecStream()
                s.next_in, s.avail_in, s.next_out, s.avail_out = a.ctypes.data, a.size, out.ctypes.data, n
                s.bits_per_sample, s.block_size, s.rsi, s.flags = bps, bs, rsi, flags
                t2 = time.perf_counter()
                rc = api.library().aec_buffer_decode(C.byref(s))
                best = min(best, time.perf_counter() - t2)
            assert rc == 0 and s.total_out == n and np.array_equal(out, data), name
            t2 = time.perf_counter()
            rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, flags, n)
            wrapped = time.perf_counter() - t2
            assert rc == 0 and dec == data.tobytes(), name
            line = (f"{bps:2d}-bit block {bs:2d} rsi {rsi} {name:20s} {args.size_mib} MiB -> {len(enc):10d} B: encode {1e3 * (t1 - t0):8.2f} ms  "
                    f"decode {1e3 * best:8.2f} ms = {n / best / 1e9:6.2f} GB/s (Python wrapper, fresh buffers: {1e3 * wrapped:6.2f} ms)")
            if have_ref() and args.size_mib <= 64:
                t3 = time.perf_counter()
                ref_decode(enc, bps, bs, rsi, flags, n)
                line += f"   reference on one core {1e3 * (time.perf_counter() - t3):8.1f} ms"
            print(line, flush=True)


if __name__ == "__main__":
    main()
