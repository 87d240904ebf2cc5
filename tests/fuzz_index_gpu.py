#!/usr/bin/env python3
"""Random sweep of the bare-stream path on the GPU box (not a pytest: minutes): random parameters and data
shapes, index pass against the encoder's offset table, decode from the found offsets against the input.

    python tests/fuzz_index_gpu.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import (AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, bytes_per_sample, pack_samples,  # noqa: E402
                     random_walk_samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="run these cases alone, e.g. 22,212 (the others only draw their random numbers)")
    ap.add_argument("--time", action="store_true", help="time a second index pass per case and list the slowest shapes")
    ap.add_argument("--list", type=int, default=25, help="with --time: how many of the slowest to list")
    return run(ap.parse_args())


def run(args):
    import torch
    from libaec_amd import gpu
    rng = np.random.default_rng(args.seed)
    only = {int(x) for x in args.only.split(",") if x != ""}
    bad = 0
    slow = []
    for case in range(args.cases):
        bps = int(rng.choice([8, 10, 12, 16, 16, 16, 24, 32]))
        bs = int(rng.choice([8, 16, 16, 32, 64]))
        rsi = int(rng.choice([1, 5, 16, 64, 128, 128, 256, 1024, 4096]))
        flags = AEC_DATA_PREPROCESS if rng.random() < 0.85 else 0
        if rng.random() < 0.3:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.3 and bps > 1:
            flags |= AEC_DATA_SIGNED
        nb = bytes_per_sample(bps, flags)
        target = int(rng.choice([1, 4, 16, 48])) << 20
        n = max(bs * rsi * 3, target // nb)
        n -= n % bs if rng.random() < 0.7 else 0
        scale = float(rng.choice([0.3, 1.0, 2.0, 8.0, 60.0, 400.0]))
        # tile a random walk so that generating stays cheap
        base_n = min(n, 1 << 20)
        vals = random_walk_samples(rng, base_n, bps, flags, scale=scale, zero_frac=float(rng.choice([0.0, 0.1, 0.5])),
                                   jump_frac=float(rng.choice([0.0, 0.001, 0.02])))
        base = pack_samples(vals, bps, flags)
        data = np.tile(base, (n * nb + base.size - 1) // base.size)[: n * nb].copy()
        if rng.random() < 0.4:                                   # incompressible stretches
            for _ in range(int(rng.integers(1, 12))):
                o = int(rng.integers(0, max(1, data.size - 70000)))
                ln = int(rng.integers(1000, 70000))
                o -= o % nb
                cnt = min(ln, data.size - o) // nb
                lo, hi = (-(1 << (bps - 1)), 1 << (bps - 1)) if flags & AEC_DATA_SIGNED else (0, 1 << bps)
                data[o:o + cnt * nb] = pack_samples(rng.integers(lo, hi, cnt), bps, flags)    # (valid samples only)
        if only and case not in only:
            continue
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        d_out, nbytes, tb, _, d_off = codec.encode(d_in)
        nr, nblk = codec.rsi_count(data.size), codec.block_count(data.size)
        d_idx = torch.zeros(nr + 2, dtype=torch.int64, device=d_in.device)
        d_res = torch.zeros(40, dtype=torch.uint8, device=d_in.device)
        codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
        whole = nblk // rsi
        why = ""
        ok = int(res[0]) in (nr, whole)
        if not ok:
            why = f"n_rsi {int(res[0])} want {nr} or {whole}; tail_blocks {int(res[1])} end_bit {int(res[2])} of {tb} status/pad {res[3]:#x}"
        elif not bool(torch.equal(d_idx[:whole], d_off[:whole])):
            ok = False
            neq = (d_idx[:whole] != d_off[:whole]).nonzero()
            first = int(neq[0])
            why = f"{neq.numel()} of {whole} offsets differ, first at RSI {first}: {int(d_idx[first])} want {int(d_off[first])}"
        if ok:
            # stream and decoded bytes against the oracle (the reference's own behaviour, also where a container holds
            # more bits than a sample: it does not give the input back there)
            from helpers import oracle_decode, oracle_encode
            rc, enc, *_ = oracle_encode(data, bps, bs, rsi, flags)
            if enc != d_out[:nbytes].cpu().numpy().tobytes():
                ok, why = False, "stream differs from the oracle's"
            else:
                d_dec, status = codec.decode(d_out, nbytes, d_idx, nr, nblk)      # (from the offsets the index pass found)
                rc2, dec_o, _ = oracle_decode(enc, bps, bs, rsi, flags, nblk * bs * nb)
                if status != 0 or d_dec.cpu().numpy().tobytes() != dec_o:
                    ok, why = False, f"decode differs from the oracle's (status {status})"
        ms = 0.0
        if args.time and ok:                     # (a second, warm index pass: which shapes are slow?)
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            scheme = gpu.INDEX_SCHEMES[gpu.index_scheme(bps, bs, rsi, flags, nbytes, nbytes * 8 // max(nr, 1))]
            slow.append((data.size / ms / 1e6, f"case {case}: bps {bps} bs {bs} rsi {rsi} flags {flags} {data.size >> 20} MiB scale {scale} "
                                                 f"ratio {data.size / nbytes:.2f}: index {ms:.2f} ms = {data.size / ms / 1e6:.2f} GB/s [{scheme}]"))
        print(f"case {case}: bps {bps} bs {bs} rsi {rsi} flags {flags} n {n} scale {scale} ratio "
              f"{data.size / nbytes:.2f}: {'ok' if ok else 'MISMATCH ' + why}", flush=True)
        bad += 0 if ok else 1
        del d_in, d_out, d_off, d_idx
    if args.time:
        print("slowest index passes (decoded bytes per second of index time):")
        for _, line in sorted(slow)[:getattr(args, 'list', 25)]:
            print("  " + line)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
