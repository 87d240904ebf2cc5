#!/usr/bin/env python3
"""One-shot aec_buffer_decode of streams with SHORT RSIs (run on the GPU box): what a narrow SZIP scan line makes
of a chunk -- RSIs of 1 .. 32 blocks, where no table scheme of the index pass applies and the phase-locked chains
(aec_idx.hip: launch_index_locked) find the RSI starts.  Product against the compiled reference on one core.

    python tests/bench_short_rsi.py [--size-mib 16]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-mib", type=int, default=16)
    ap.add_argument("--size-kib", type=int, default=0, help="a small chunk instead (what one SZIP call per chunk decodes): best of 50 calls")
    ap.add_argument("--rsi", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32, 64, 128])
    ap.add_argument("--edges", action="store_true", help="the shapes at the schemes' thresholds instead (ADVICE round 4): rsi 33 .. 63, "
                    "no preprocessor")
    args = ap.parse_args()
    import torch  # noqa: F401
    from helpers import have_ref, ref_decode
    from libaec_amd import api
    from test_gpu_parity import gen
    n = (args.size_kib << 10) if args.size_kib else (args.size_mib << 20)
    size = f"{args.size_kib} KiB" if args.size_kib else f"{args.size_mib} MiB"
    PP = api.AEC_DATA_PREPROCESS
    plan = [(kind, bps, bs, rsi, PP) for kind, bps, bs in ((2, 8, 8), (0, 16, 16)) for rsi in args.rsi]
    if args.edges:
        plan = [(kind, bps, bs, rsi, fl) for kind, bps, bs in ((2, 8, 8), (0, 16, 16))
                for rsi, fl in ((33, PP), (40, PP), (48, PP), (63, PP), (4, 0), (16, 0), (128, 0), (256, 0))]
        # (AEC_PAD_RSI is not here: the reference does not decode what its encoder makes of these inputs with that flag --
        # AEC_DATA_ERROR, and the product returns the same; tests/test_gpu_parity.py has the parity of that)
    cache = {}
    for kind, bps, bs, rsi, flags in plan:
        if kind not in cache:
            cache[kind] = gen(kind, n)
        data = cache[kind]
        for _ in (0,):
            rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
            assert rc == 0
            best = 1e9
            for _ in range(50 if args.size_kib else 3):
                t0 = time.perf_counter()
                rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, flags, n)
                best = min(best, time.perf_counter() - t0)
            assert rc == 0 and np.array_equal(np.frombuffer(dec, dtype=np.uint8), data)
            ref = ""
            if have_ref():
                t0 = time.perf_counter()
                rc_r, dec_r = ref_decode(enc, bps, bs, rsi, flags, n)
                ref = f"   reference on one core {(time.perf_counter() - t0) * 1e3:8.1f} ms"
                assert dec_r == dec
            what = "" if flags == PP else " no preprocessor"
            print(f"{bps}-bit block {bs} rsi {rsi:3d}{what}, {size} (stream {len(enc) >> 10} KiB): aec_buffer_decode "
                  f"{best * 1e3:8.2f} ms = {n / best / 1e9:6.2f} GB/s{ref}", flush=True)


if __name__ == "__main__":
    main()
