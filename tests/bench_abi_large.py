#!/usr/bin/env python3
"""Repeated large one-shot ABI calls on pinned host buffers (what abi_end_to_end in bench.py measures once):
    python tests/bench_abi_large.py [--config c2] [--size-mib 256] [--reps 4]
prints the time of every call, so one-time costs (module load, workspace allocation) show up in the first."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--size-mib", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--pageable", action="store_true", help="ordinary host memory instead of pinned buffers")
    args = ap.parse_args()
    import torch
    from libaec_amd import api
    lib = api.library()
    name, kind, bps, bs, rsi, flags = bench.CONFIGS[args.config]
    n = args.size_mib << 20
    host_np = bench.typical_tiled(n) if kind < 0 else bench.generate(kind, n, 0, 8)
    pin = (lambda t: t) if args.pageable else (lambda t: t.pin_memory())
    t_host = pin(torch.from_numpy(np.asarray(host_np).view(np.uint8)[:n].copy()))
    t_enc = pin(torch.zeros(n + n // 8 + (1 << 20), dtype=torch.uint8))
    t_dec = pin(torch.zeros(n, dtype=torch.uint8))
    host, enc, dec = t_host.numpy(), t_enc.numpy(), t_dec.numpy()

    def call(fn, src, src_len, dst):
        st = api.AecStream()
        st.next_in, st.avail_in = src.ctypes.data, src_len
        st.next_out, st.avail_out = dst.ctypes.data, dst.size
        st.bits_per_sample, st.block_size, st.rsi, st.flags = bps, bs, rsi, flags
        t0 = time.perf_counter()
        rc = getattr(lib, fn)(C.byref(st))
        dt = time.perf_counter() - t0
        assert rc == 0, (fn, rc)
        return st.total_out, dt

    for i in range(args.reps):
        clen, te = call("aec_buffer_encode", host, n, enc)
        dlen, td = call("aec_buffer_decode", enc, clen, dec)
        assert dlen == n and np.array_equal(dec, host)
        print(f"{args.config} {args.size_mib} MiB {'pageable' if args.pageable else 'pinned'} call {i}: encode {te * 1e3:8.2f} ms ({n / te / 1e9:6.2f} GB/s)   "
              f"decode {td * 1e3:8.2f} ms ({n / td / 1e9:6.2f} GB/s)   stream {clen} B", flush=True)


if __name__ == "__main__":
    main()
