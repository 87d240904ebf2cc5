#!/usr/bin/env python3
"""Chunked SZIP throughput (run on the GPU box): an HDF5-style dataset of 1 MiB chunks of 8-bit pixels
(BASELINE config 5 shape) through (a) one SZ_BufftoBuff call per chunk, (b) SZ_BatchCompress /
SZ_BatchDecompress with all chunks in one call, (c) the reference shim on one core.  Host buffers in and
out (pageable): these are PCIe-inclusive rates."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    run(64, 1 << 20)
    run(1024, 64 << 10)


def run(n, chunk):
    import torch  # noqa: F401
    from helpers import REF_SO, have_ref
    from libaec_amd import szip
    from test_gpu_parity import gen
    opts = szip.SZ_NN_OPTION_MASK | szip.SZ_RAW_OPTION_MASK
    data = gen(2, n * chunk)
    chunks = [data[i * chunk:(i + 1) * chunk] for i in range(n)]
    sizes = [chunk * 2] * n
    szip.compress(chunks[0], chunk * 2, opts, 8, 8, 1024)                      # warm up

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            best = min(best, time.perf_counter() - t0)
        return best, out

    t, comp = timed(lambda: [szip.compress(c, chunk * 2, opts, 8, 8, 1024)[1] for c in chunks])
    print(f"compress   {n} x {chunk >> 10} KiB, one call per chunk : {t * 1e3:8.2f} ms  {n * chunk / t / 1e9:7.2f} GB/s")
    t, (rc, comp_b, st) = timed(lambda: szip.compress_batch(chunks, sizes, opts, 8, 8, 1024))
    assert rc == 0 and comp_b == comp
    print(f"compress   {n} x {chunk >> 10} KiB, SZ_BatchCompress       : {t * 1e3:8.2f} ms  {n * chunk / t / 1e9:7.2f} GB/s")
    t, dec = timed(lambda: [szip.decompress(c, chunk, opts, 8, 8, 1024)[1] for c in comp])
    assert dec == [c.tobytes() for c in chunks]
    print(f"decompress {n} x {chunk >> 10} KiB, one call per chunk : {t * 1e3:8.2f} ms  {n * chunk / t / 1e9:7.2f} GB/s")
    t, (rc, dec_b, st) = timed(lambda: szip.decompress_batch(comp, [chunk] * n, opts, 8, 8, 1024))
    assert rc == 0 and dec_b == dec
    print(f"decompress {n} x {chunk >> 10} KiB, SZ_BatchDecompress     : {t * 1e3:8.2f} ms  {n * chunk / t / 1e9:7.2f} GB/s")
    if have_ref():
        ref = szip.bind(C.CDLL(REF_SO))
        t, _ = timed(lambda: [szip.compress(c, chunk * 2, opts, 8, 8, 1024, lib=ref)[1] for c in chunks[:8]], 1)
        print(f"compress   reference shim, one core          : {t / 8 * n * 1e3:8.2f} ms  {8 * chunk / t / 1e9:7.2f} GB/s")
        t, _ = timed(lambda: [szip.decompress(c, chunk, opts, 8, 8, 1024, lib=ref)[1] for c in comp[:8]], 1)
        print(f"decompress reference shim, one core          : {t / 8 * n * 1e3:8.2f} ms  {8 * chunk / t / 1e9:7.2f} GB/s")


if __name__ == "__main__":
    main()
