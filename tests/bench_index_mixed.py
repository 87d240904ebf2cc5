#!/usr/bin/env python3
"""Index pass on a low-entropy stream with incompressible RSIs sprinkled in (run on the GPU box): what the
fallbacks cost when some RSIs do not fit the look-ahead of the window tables.

    python tests/bench_index_mixed.py [--size-mib 256] [--noise-permille 0 1 10 50]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-mib", type=int, default=1024)
    ap.add_argument("--noise-permille", type=int, nargs="+", default=[0, 1, 10, 50])
    ap.add_argument("--noise-bits", type=int, default=16,
                    help="random bits per sample of the sprinkled RSIs (16: incompressible; 13: long but compressible)")
    args = ap.parse_args()
    import torch
    from libaec_amd import gpu
    name, kind, bps, bs, rsi, flags = bench.CONFIGS["c2"]
    dev = torch.device("cuda", 0)
    n = args.size_mib << 20
    base = bench.generate(kind, n, 0, os.cpu_count() or 8)
    rsi_bytes = rsi * bs * 2
    nr = n // rsi_bytes
    rng = np.random.default_rng(1)
    for pm in args.noise_permille:
        host = base.copy()
        pick = rng.choice(nr, size=nr * pm // 1000, replace=False)
        for r in pick:
            host[r * rsi_bytes:(r + 1) * rsi_bytes] = rng.integers(0, 1 << args.noise_bits, rsi_bytes // 2, dtype=np.uint16).view(np.uint8)
        if os.environ.get("MIX_DEBUG"):
            print("noise RSIs:", " ".join(str(int(r)) for r in sorted(pick)[:3000]))
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(host).to(dev)
        d_out, cbytes, bits, _, d_off = codec.encode(d_in)
        d_idx = torch.zeros(nr + 2, dtype=torch.int64, device=dev)
        d_res = torch.zeros(40, dtype=torch.uint8, device=dev)
        best = 1e9
        for rep in range(3):
            d_idx.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codec.index_async(d_out, cbytes, 0, d_idx, nr, d_res)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ok = bool(torch.equal(d_idx[:nr], d_off[:nr]))
        print(f"c2 {args.size_mib} MiB, {pm} per mille RSIs of {args.noise_bits} random bits ({len(pick)}): stream {cbytes >> 20} MiB, "
              f"index {best * 1e3:.2f} ms = {n / best / 1e9:.2f} GB/s decoded-equivalent, offsets {'OK' if ok else 'MISMATCH'}")


if __name__ == "__main__":
    main()
