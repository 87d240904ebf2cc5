#!/usr/bin/env python3
"""Index-pass speed on bare streams (run on the GPU box): encode synthetic data, drop the offset
table, find the RSI starts again with aec_gpu_index_async and compare with the encoder's table.

    python tests/bench_index.py [--config c2|c5|c3|typical] [--size-mib 64 1024]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def held(codec):
    """aec_gpu_held_bytes: device memory the context holds for its own purposes (index tables, encoder workspace)"""
    import ctypes as C
    codec.lib.aec_gpu_held_bytes.restype = C.c_size_t
    return int(codec.lib.aec_gpu_held_bytes(codec.ctx))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--size-mib", type=int, nargs="+", default=[1, 64, 1024])
    ap.add_argument("--decode", action="store_true",
                    help="also decode: aec_gpu_index_segments_async + aec_gpu_decode_bare_async, timed apart")
    args = ap.parse_args()
    import torch
    from libaec_amd import gpu
    name, kind, bps, bs, rsi, flags = bench.CONFIGS[args.config]
    bench.BPS, bench.BS, bench.RSI, bench.FLAGS = bps, bs, rsi, flags
    dev = torch.device("cuda", 0)
    for mib in args.size_mib:
        n = mib << 20
        host = bench.typical_tiled(n) if kind < 0 else bench.generate(kind, n, 0, os.cpu_count() or 8)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(host).to(dev)
        d_out, cbytes, bits, _, d_off = codec.encode(d_in)
        nr = codec.rsi_count(n)
        d_idx = torch.zeros(nr + 2, dtype=torch.int64, device=dev)
        d_res = torch.zeros(40, dtype=torch.uint8, device=dev)
        for rep in range(3):
            d_idx.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codec.index_async(d_out, cbytes, 0, d_idx, nr + 1, d_res)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        if args.decode:
            d_sb = torch.zeros((nr + 2) * codec.segments_per_rsi(), dtype=torch.int64, device=dev)
            d_dec = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)
            d_dres = torch.zeros(40, dtype=torch.uint8, device=dev)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for rep in range(3):
                e[0].record()
                codec.index_segments_async(d_out, cbytes, 0, d_idx, d_sb, nr + 1, d_res)
                e[1].record()
                codec.decode_bare_async(d_out, cbytes, d_idx, d_sb, nr, codec.block_count(n), None, d_dec, d_dres)
                e[2].record()
                torch.cuda.synchronize()
            assert torch.equal(d_dec[:n], d_in), "bare decode differs"
            print(f"{args.config} {mib} MiB: index + segment starts {e[0].elapsed_time(e[1]):.3f} ms, decode "
                  f"{e[1].elapsed_time(e[2]):.3f} ms", flush=True)
        res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
        ok = bool(torch.equal(d_idx[:nr], d_off[:nr])) and int(res["n_rsi"]) == nr and int(res["end_bit"]) == bits
        print(f"{args.config} {mib} MiB: index {dt * 1e3:.3f} ms = {n / dt / 1e9:.3f} GB/s decoded-equivalent "
              f"({cbytes / dt / 1e9:.3f} GB/s of stream), n_rsi {int(res['n_rsi'])}/{nr}, "
              f"status {int(res['status'])}, offsets {'OK' if ok else 'MISMATCH'}; the context holds "
              f"{held(codec) / 2 ** 20:.0f} MiB (index tables and workspaces)", flush=True)
        assert ok


if __name__ == "__main__":
    main()
