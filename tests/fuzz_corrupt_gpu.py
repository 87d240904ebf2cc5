#!/usr/bin/env python3
"""Corrupted streams through aec_buffer_decode on the GPU box: random bit flips, cuts and garbage tails in valid
streams.  Nothing may hang or fault; where the oracle (= the reference's behaviour) decodes the damaged stream
without an error, the product must give the same bytes, and where it reports AEC_DATA_ERROR the product may not
report success with different bytes.

    python tests/fuzz_corrupt_gpu.py [--cases 150] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import (AEC_DATA_ERROR, AEC_DATA_PREPROCESS, AEC_OK, bytes_per_sample, have_ref, oracle_decode,  # noqa: E402
                     oracle_encode, pack_samples, random_walk_samples, ref_decode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    return run(ap.parse_args())


def run(args):
    import torch  # noqa: F401
    from libaec_amd import api
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        bps = int(rng.choice([8, 16, 16, 32]))
        bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 16, 128, 1024]))
        flags = AEC_DATA_PREPROCESS
        nb = bytes_per_sample(bps, flags)
        n = int(rng.choice([2000, 40000, 300000, 2000000]))
        vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.5, 3.0, 50.0])), zero_frac=0.2, jump_frac=0.002)
        data = pack_samples(vals, bps, flags)
        rc, enc, *_ = oracle_encode(data, bps, bs, rsi, flags)
        enc = bytearray(enc)
        kind = int(rng.integers(0, 4))
        out_size = ((n + bs - 1) // bs) * bs * nb
        if kind == 3:
            # the WHOLE stream + a garbage tail, output of exactly the decoded size: the reference fills the output and
            # looks no further (decode.c:797-831; a zero run that overruns its RSI behind a full output is no error,
            # :542-544) -- compared with the compiled reference itself where it travelled
            enc = bytes(enc) + bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8).tolist())
            rc_o, dec_o = ref_decode(enc, bps, bs, rsi, flags, out_size) if have_ref() else \
                oracle_decode(enc, bps, bs, rsi, flags, out_size)[:2]
            rc_p, dec_p = api.aec_buffer_decode(enc, bps, bs, rsi, flags, out_size)
            if (rc_p, dec_p) != (rc_o, dec_o):
                bad += 1
                print(f"case {case}: bps {bps} bs {bs} rsi {rsi} n {n} kind 3 (tail of {len(enc)} bytes): reference rc {rc_o} "
                      f"{len(dec_o)} bytes, product rc {rc_p} {len(dec_p)} bytes, equal {dec_p == dec_o}", flush=True)
            continue
        if kind == 0:                                   # bit flips
            for _ in range(int(rng.integers(1, 6))):
                enc[int(rng.integers(0, len(enc)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # cut + garbage tail
            cut = int(rng.integers(1, len(enc)))
            enc = enc[:cut] + bytes(rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8).tolist())
        else:                                           # a stretch overwritten
            o = int(rng.integers(0, len(enc)))
            ln = int(rng.integers(1, 200))
            enc[o:o + ln] = bytes(rng.integers(0, 256, min(ln, len(enc) - o), dtype=np.uint8).tolist())
        enc = bytes(enc)
        rc_o, dec_o, _ = oracle_decode(enc, bps, bs, rsi, flags, out_size)
        rc_p, dec_p = api.aec_buffer_decode(enc, bps, bs, rsi, flags, out_size)
        ok = rc_p in (AEC_OK, AEC_DATA_ERROR)
        if ok and rc_o == AEC_OK:
            ok = rc_p == AEC_OK and dec_p == dec_o
        elif ok and rc_p != AEC_DATA_ERROR:
            # The oracle answers as the reference does with AMPLE room.  A zero run that overruns its RSI is refused by
            # the reference only when the call's room holds the whole run (decode.c:542-544; else m_zero_output,
            # :504-516, fills what room there is and the call ends OK), and an error behind a full output is never
            # looked at (:797-831): with exactly this room the compiled reference itself is the judge.
            ok = False
            if have_ref():
                rc_r, dec_r = ref_decode(enc, bps, bs, rsi, flags, out_size)
                ok = (rc_p, dec_p) == (rc_r, dec_r)
                if ok:
                    print(f"case {case}: oracle rc {rc_o}, reference with this room rc {rc_r} {len(dec_r)} bytes = product", flush=True)
        if not ok:
            bad += 1
            print(f"case {case}: bps {bps} bs {bs} rsi {rsi} n {n} kind {kind}: oracle rc {rc_o} {len(dec_o)} bytes, "
                  f"product rc {rc_p} {len(dec_p)} bytes, equal {dec_p == dec_o}", flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
