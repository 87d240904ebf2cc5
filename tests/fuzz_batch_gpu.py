#!/usr/bin/env python3
"""Random sweep of the batch entry points on the GPU box (aec_buffer_encode_batch / aec_buffer_decode_batch): random
parameters, numbers and shapes of chunks -- equal chunks of whole RSIs (one launch set for all), ragged ones, tiny
and large, some with too small an output buffer, some streams corrupted -- every chunk against the oracle.

    python tests/fuzz_batch_gpu.py [--cases 40] [--seed 1]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import (AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, AEC_OK, AEC_STREAM_ERROR, bytes_per_sample,  # noqa: E402
                     oracle_decode, oracle_encode, pack_samples, random_walk_samples)


def batch(lib, name, params, srcs, caps):
    from libaec_amd import api
    fn = getattr(lib, name)
    fn.restype = C.c_int
    n = len(srcs)
    st = api.AecStream()
    st.bits_per_sample, st.block_size, st.rsi, st.flags = params
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in caps]
    src = (C.c_void_p * n)(*[a.ctypes.data for a in srcs])
    src_len = (C.c_size_t * n)(*[a.size for a in srcs])
    dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    dst_len = (C.c_size_t * n)(*caps)
    status = (C.c_int * n)()
    rc = fn(C.byref(st), C.c_size_t(n), src, src_len, dst, dst_len, status)
    return rc, [o[:dst_len[i]] for i, o in enumerate(outs)], list(status)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    return run(ap.parse_args())


def run(args):
    import torch  # noqa: F401
    from libaec_amd import api
    lib = api.library()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        bps = int(rng.choice([8, 8, 12, 16, 16, 32]))
        bs = int(rng.choice([8, 16, 16, 32, 64]))
        rsi = int(rng.choice([1, 8, 64, 128, 128, 512]))
        flags = AEC_DATA_PREPROCESS if rng.random() < 0.85 else 0
        if rng.random() < 0.3:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.3:
            flags |= AEC_DATA_SIGNED
        nb = bytes_per_sample(bps, flags)
        rsi_bytes = rsi * bs * nb
        n = int(rng.choice([1, 2, 5, 17, 64, 100, 300]))
        shape = int(rng.integers(0, 4))            # 0 equal whole RSIs, 1 equal ragged, 2 random sizes, 3 mixed tiny / large
        if shape == 0:
            sizes = [rsi_bytes * int(rng.integers(1, max(2, min(2048 * 64 // max(rsi, 1), (256 << 10) // rsi_bytes + 2))))] * n
        elif shape == 1:
            sizes = [int(rng.integers(nb, 200000)) // nb * nb] * n
        elif shape == 2:
            sizes = [int(rng.integers(nb, 300000)) // nb * nb for _ in range(n)]
        else:
            sizes = [int(rng.choice([nb * 3, rsi_bytes, 70000 // nb * nb, (1 << 20) // nb * nb])) for _ in range(n)]
        total = sum(sizes)
        if total > (96 << 20):
            sizes = [max(nb, s * (96 << 20) // total // nb * nb) for s in sizes]
        scale = float(rng.choice([0.5, 2.0, 30.0]))
        vals = random_walk_samples(rng, 1 << 18, bps, flags, scale=scale, zero_frac=0.1, jump_frac=0.002)
        base = pack_samples(vals, bps, flags)
        chunks = []
        for s in sizes:
            o = int(rng.integers(0, base.size)) // nb * nb
            chunks.append(np.ascontiguousarray(np.resize(np.roll(base, -o), s)))
        want = [oracle_encode(c, bps, bs, rsi, flags)[1] for c in chunks]
        caps = [len(w) + 64 for w in want]
        small = int(rng.integers(0, n)) if rng.random() < 0.3 else -1
        if small >= 0:
            caps[small] = max(1, len(want[small]) // 2)
        rc, got, st = batch(lib, "aec_buffer_encode_batch", (bps, bs, rsi, flags), chunks, caps)
        ok, why = True, ""
        for i in range(n):
            if i == small:
                if st[i] != AEC_STREAM_ERROR or got[i].tobytes() != want[i][:caps[i]]:
                    ok, why = False, f"chunk {i} (output too small): status {st[i]}, {got[i].size} bytes"
            elif st[i] != AEC_OK or got[i].tobytes() != want[i]:
                ok, why = False, f"encode chunk {i} of {n}: status {st[i]}, {got[i].size} bytes, want {len(want[i])}"
            if not ok:
                break
        if ok:
            streams = [np.frombuffer(w, dtype=np.uint8).copy() for w in want]
            nblk = [(s // nb + bs - 1) // bs for s in sizes]
            dec_want = [oracle_decode(w, bps, bs, rsi, flags, k * bs * nb)[1] for w, k in zip(want, nblk)]
            rc, dec, st = batch(lib, "aec_buffer_decode_batch", (bps, bs, rsi, flags), streams, [k * bs * nb for k in nblk])
            for i in range(n):
                if st[i] != AEC_OK or dec[i].tobytes() != dec_want[i]:
                    ok, why = False, f"decode chunk {i} of {n}: status {st[i]}, {dec[i].size} bytes, want {len(dec_want[i])}"
                    break
        print(f"case {case}: bps {bps} bs {bs} rsi {rsi} flags {flags} n {n} shape {shape} sizes {sizes[0]}.. : "
              f"{'ok' if ok else 'MISMATCH ' + why}", flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
