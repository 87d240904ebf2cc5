#!/usr/bin/env python3
"""Latency of small one-shot ABI calls (the HDF5 / SZIP chunk pattern): aec_buffer_encode and
aec_buffer_decode of one chunk at a time, product library vs the reference on one core.

    python tests/bench_abi_small.py [--chunk-kib 64 1024] [--config c5]
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
import bench  # noqa: E402
import helpers  # noqa: E402


def one_shot(lib, st_type, fn, src, n_in, dst, params):
    st = st_type()
    st.next_in, st.avail_in = src.ctypes.data, n_in
    st.next_out, st.avail_out = dst.ctypes.data, dst.size
    st.bits_per_sample, st.block_size, st.rsi, st.flags = params
    rc = getattr(lib, fn)(C.byref(st))
    assert rc == 0, (fn, rc)
    return st.total_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5")
    ap.add_argument("--chunk-kib", type=int, nargs="+", default=[64, 1024, 16384])
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from libaec_amd import api
    name, kind, bps, bs, rsi, flags = bench.CONFIGS[args.config]
    params = (bps, bs, rsi, flags)
    libs = [("product", api.library(), api.AecStream)]
    if helpers.have_ref():
        libs.append(("reference", helpers.ref_lib(), helpers.AecStream))
    for kib in args.chunk_kib:
        n = kib << 10
        host = bench.generate(kind, n, 0, 4)
        enc = np.zeros(n + n // 2 + 4096, dtype=np.uint8)
        dec = np.zeros(n, dtype=np.uint8)
        for who, lib, st_type in libs:
            clen = one_shot(lib, st_type, "aec_buffer_encode", host, n, enc, params)       # warm
            one_shot(lib, st_type, "aec_buffer_decode", enc, clen, dec, params)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                clen = one_shot(lib, st_type, "aec_buffer_encode", host, n, enc, params)
            t1 = time.perf_counter()
            for _ in range(args.reps):
                one_shot(lib, st_type, "aec_buffer_decode", enc, clen, dec, params)
            t2 = time.perf_counter()
            assert np.array_equal(dec, host)
            print(f"{args.config} chunk {kib} KiB {who:9s}: encode {(t1 - t0) / args.reps * 1e3:8.3f} ms "
                  f"({n / ((t1 - t0) / args.reps) / 1e9:6.2f} GB/s)   decode {(t2 - t1) / args.reps * 1e3:8.3f} ms "
                  f"({n / ((t2 - t1) / args.reps) / 1e9:6.2f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
