#!/usr/bin/env python3
"""Every index scheme that may legally take a stream, forced on the SAME streams (run on the GPU box with the tuning library:
tests/test_gpu_parity.py::test_every_index_scheme_on_the_same_streams).

launch_index (libaec_amd/csrc/aec_idx.hip) takes the first scheme that applies -- every bit parsed, regions, phase-locked
chains (the 64 agreeing chains or entries by plausibility), window tables, trunk -- by thresholds that were set by
measurement; a shape is otherwise only ever tested through the scheme today's thresholds pick.  Here the switches of the
tuning build (aec_tune.h: AEC_IDX_SMALL, AEC_IDX_REGIONS, AEC_IDX_LOCK, AEC_IDX_LOCK_P, AEC_IDX_NO_SPARSE) take the schemes
away one after the other, aec_gpu_index_scheme says which one is left, and for each: the RSI starts must be the oracle
encoder's table, the record the end of the stream, the decoded bytes the oracle's -- for the whole stream, a stream cut
short, a stream with bytes behind the caller's bound, and a walk that resumes inside an RSI.

    AEC_AMD_LIB=libaec_amd/lib/tuning/libaec.so.0 python tests/cross_scheme.py [--quick]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

PP, MSB, SGN = H.AEC_DATA_PREPROCESS, H.AEC_DATA_MSB, H.AEC_DATA_SIGNED
SCHEMES = ("serial walk", "phase-locked chains", "window tables", "trunk", "every bit parsed", "regions")
# what takes a scheme out of the dispatch (tuning build), in the order launch_index tries them
OFF = {4: {"AEC_IDX_SMALL": "0"}, 5: {"AEC_IDX_REGIONS": "0"}, 1: {"AEC_IDX_LOCK": "0", "AEC_IDX_LOCK_P": "0"},
       2: {"AEC_IDX_NO_SPARSE": "1"}}
ORDER = (4, 5, 1, 2, 3)
KNOBS = sorted({k for v in OFF.values() for k in v} | {"AEC_IDX_REGIONS_MIN"})


def smooth(rng, n, bps, signed, scale, zero_frac=0.05, jump_frac=0.0005):
    """a bounded walk with constant stretches (zero blocks, rest-of-segment runs) and a few jumps, vectorised"""
    steps = np.rint(rng.standard_normal(n) * scale * rng.choice([0.3, 1, 6], size=n, p=[0.5, 0.4, 0.1])).astype(np.int64)
    nz = max(1, int(n * zero_frac / 300))
    for s in rng.integers(0, n, nz):
        steps[s:s + int(rng.integers(10, 2000))] = 0
    jumps = rng.random(n) < jump_frac
    steps[jumps] = rng.integers(-(1 << (bps - 2)), 1 << (bps - 2), int(jumps.sum()))
    span = (1 << bps) - 1
    x = np.cumsum(steps) + (span >> 1)
    x = np.abs((x % (2 * span)) - span)                  # reflected into [0, span]
    if signed:
        x = x - (1 << (bps - 1))
    return x


PAD = H.AEC_PAD_RSI


def make_stream(rng, bps, bs, rsi, flags, nbytes, scale):
    nb = H.bytes_per_sample(bps, flags)
    n = nbytes // nb // bs * bs                          # whole blocks
    data = H.pack_samples(smooth(rng, n, bps, bool(flags & SGN), scale), bps, flags)
    if flags & PAD:
        # AEC_PAD_RSI is the DECODER's flag (reference decode.c:407-408; the encoder's padding is dead code, encode.c:499-505):
        # a stream whose RSIs begin on bytes is the RSIs coded one by one, each padded to a byte
        rb = bs * rsi * nb
        parts, traces, offs, pos = [], [], [], 0
        for i in range(0, data.size, rb):
            rc, e, t, _, b = H.oracle_encode(data[i:i + rb], bps, bs, rsi, flags & ~PAD, want_trace=True)
            assert rc == H.AEC_OK
            t = t.copy()
            # (the padding counts as the last coded data set's -- the last block may lie inside a run of zero blocks)
            t["bits"][np.flatnonzero(t["bits"])[-1]] += (-b) % 8
            parts.append(e)
            traces.append(t)
            offs.append(pos)
            pos += len(e) * 8
        tr = np.concatenate(traces)
        # (the walk ends behind the last coded data set -- on the next byte, if that completed an RSI: decode.c:407-408)
        last_bits = pos if (n // bs) % rsi == 0 else pos - (-b) % 8
        return np.ascontiguousarray(data), b"".join(parts), tr, np.array(offs, dtype=np.uint64), last_bits
    rc, enc, tr, offs, bits = H.oracle_encode(data, bps, bs, rsi, flags, want_trace=True)
    assert rc == H.AEC_OK
    return np.ascontiguousarray(data), enc, tr, offs, bits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import torch
    from libaec_amd import gpu
    assert "tuning" in os.environ.get("AEC_AMD_LIB", ""), "the switches exist in the tuning build only"
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(606)
    KiB, MiB = 1 << 10, 1 << 20
    shapes = []
    for rsi in (1, 16, 44, 45, 64, 65, 256, 257, 4096):
        for size in (100 * KiB, 600 * KiB, 3 * MiB, 20 * MiB):
            if args.quick and size > 3 * MiB:
                continue
            if rsi in (44, 65, 257) and size in (600 * KiB,):
                continue                                     # (thin the grid: ~30 streams)
            if rsi in (16, 45, 64) and size == 20 * MiB:
                continue
            bps, bs = ((8, 8) if (rsi + size // KiB) % 3 == 0 else (16, 16) if (rsi + size // KiB) % 3 == 1 else (32, 32))
            if rsi == 4096 and size < 3 * MiB:
                bps, bs = 8, 8                               # (a few RSIs at least)
            flags = PP | (MSB if rsi % 2 else 0) | (SGN if bps == 32 else 0)
            shapes.append((bps, bs, rsi, flags, size, 2.0 if bps > 8 else 1.2))
    # without the preprocessor, and the benchmark's own shapes at a size where the region index may be forced
    shapes += [(16, 16, 16, 0, 600 * KiB, 30.0), (8, 8, 128, 0, 3 * MiB, 3.0), (16, 16, 128, PP, 20 * MiB, 2.0),
               (8, 8, 128, PP, 20 * MiB, 1.2), (16, 64, 256, PP | MSB, 20 * MiB, 40.0)]
    # AEC_PAD_RSI: every RSI begins on a byte (the every-bit scheme from round 6 on, the trunk, the serial walker)
    shapes += [(16, 16, 8, PP | PAD, 100 * KiB, 2.0), (8, 8, 64, PP | PAD | MSB, 600 * KiB, 1.2), (16, 16, 128, PAD, 600 * KiB, 30.0)]
    ran = {s: 0 for s in range(6)}
    for (bps, bs, rsi, flags, size, scale) in shapes:
        data, enc, tr, offs, bits = make_stream(rng, bps, bs, rsi, flags, size, scale)
        n_rsi = len(offs)
        hint = bits // max(1, n_rsi)
        codec = gpu.Codec(bps, bs, rsi, flags)
        nb = H.bytes_per_sample(bps, flags)
        blk_bytes = bs * nb
        d_enc = torch.from_numpy(np.frombuffer(enc + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
        # coded data set boundaries for the walk that resumes inside an RSI
        cds_bits = tr["bits"].astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(cds_bits)[:-1]])
        taken = []
        disabled = {}
        for s in ORDER:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(disabled)
            os.environ["AEC_IDX_REGIONS_MIN"] = "0"           # (the region index wherever it may run at all)
            scheme = gpu.index_scheme(bps, bs, rsi, flags, len(enc), hint, 0)
            if s in OFF:
                disabled = {**disabled, **OFF[s]}
            if scheme != s or scheme in taken:
                continue
            taken.append(scheme)
            ran[scheme] += 1
            tag = f"{bps}-bit block {bs} rsi {rsi} flags {flags} {size >> 10} KiB [{SCHEMES[scheme]}]"

            def index(in_bytes, max_rsi, start=(0, 0, 0)):
                d_idx = torch.full((max_rsi + 2,), -1, dtype=torch.int64, device=dev)
                d_res = torch.zeros(40, dtype=torch.uint8, device=dev)
                if start == (0, 0, 0):
                    codec.index_async(d_enc, in_bytes, 0, d_idx, max_rsi, d_res)
                else:
                    codec.index_resume_async(d_enc, in_bytes, start[0], start[1], start[2], d_idx, max_rsi, d_res)
                torch.cuda.synchronize()
                return d_idx, d_res, d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]

            # 1. the whole stream
            d_idx, d_res, res = index(len(enc), n_rsi + 1)
            got = d_idx.cpu().numpy()[:n_rsi].astype(np.uint64)
            assert np.array_equal(got, offs), (tag, "RSI starts", int(np.argmax(got != offs)))
            last_full = (len(data) // blk_bytes) % rsi == 0
            assert int(res["status"]) == 0 and int(res["n_rsi"]) == (n_rsi if last_full else n_rsi - 1), (tag, res)
            assert int(res["end_bit"]) == bits, (tag, "end", int(res["end_bit"]), bits)
            d_out = torch.zeros(len(data) + 4096, dtype=torch.uint8, device=dev)
            d_dres = torch.zeros(40, dtype=torch.uint8, device=dev)
            codec.decode_indexed_async(d_enc, len(enc), d_idx, n_rsi + 1, d_res, d_out, d_dres)
            torch.cuda.synchronize()
            assert int(d_dres.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]["status"]) == 0, tag
            assert d_out[:len(data)].cpu().numpy().tobytes() == data.tobytes(), (tag, "decoded bytes")
            if n_rsi < 3:
                continue
            # 2. cut short: the RSI starts in front of the cut, and the record says where the input ended
            cut = int(rng.integers(int(offs[n_rsi // 2]) // 8 + 1, len(enc) - 1))
            d_idx, d_res, res = index(cut, n_rsi + 1)
            whole = int(np.searchsorted(offs, cut * 8, side="right")) - 1          # RSIs that began in front of the cut
            k = int(res["n_rsi"])
            assert int(res["status"]) <= 1 and whole - 1 <= k <= whole, (tag, "cut", k, whole)
            assert np.array_equal(d_idx.cpu().numpy()[:k + 1].astype(np.uint64), offs[:k + 1]), (tag, "cut: RSI starts")
            assert int(res["end_bit"]) <= cut * 8
            # 3. the caller's bound in front of the end of the input (bytes behind it are none of the walk's business)
            bound = max(1, n_rsi // 3)
            d_idx, d_res, res = index(len(enc), bound)
            assert int(res["n_rsi"]) == bound and int(res["end_bit"]) == int(offs[bound]), (tag, "bound", res)
            assert np.array_equal(d_idx.cpu().numpy()[:bound].astype(np.uint64), offs[:bound]), (tag, "bound: RSI starts")
            # 4. a walk that resumes inside an RSI (streaming callers): RSI number 0 is the one it resumes in
            if rsi > 1:
                r0 = n_rsi // 4
                j = int(rng.integers(1, rsi))
                blk = r0 * rsi + j
                # (the coded data set that holds block blk must begin on it: not inside a run of zero blocks)
                while blk < (r0 + 1) * rsi and cds_bits[blk] == 0:
                    blk += 1
                if blk < (r0 + 1) * rsi and blk < len(starts):
                    j = blk - r0 * rsi
                    d_idx, d_res, res = index(len(enc), n_rsi - r0 + 1, (int(starts[blk]), j, int(offs[r0])))
                    got = d_idx.cpu().numpy()[:n_rsi - r0].astype(np.uint64)
                    assert np.array_equal(got, offs[r0:]), (tag, "resumed walk", j)
                    assert int(res["end_bit"]) == bits, (tag, "resumed walk: end")
        assert taken, (bps, bs, rsi, flags, size)
        codec.close()
        print(f"{bps}-bit block {bs} rsi {rsi} flags {flags} {size >> 10} KiB ({n_rsi} RSIs of {hint} bits): "
              + ", ".join(SCHEMES[s] for s in taken), flush=True)
    print("schemes exercised:", {SCHEMES[s]: n for s, n in ran.items() if n})
    assert all(ran[s] for s in (1, 2, 3, 4, 5)), ran
    print("cross scheme ok")


if __name__ == "__main__":
    main()
