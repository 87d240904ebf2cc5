#!/usr/bin/env python3
"""The streaming interface (aec_decode_init / aec_decode / aec_decode_end and the encoder's) with random chunking on
the GPU box, the SAME sequence of calls driven against the reference library (oracle/_ref) and the product: the bytes
that have come out when the input is exhausted and the calls bring nothing more, and the last return code, must be the
same -- for valid streams and for damaged ones.  (Per-call amounts may differ: the product hands out in batches.)

    python tests/fuzz_stream_gpu.py [--cases 60] [--seed 1]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import (AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, AEC_FLUSH, AEC_NO_FLUSH, AEC_OK, AecStream,  # noqa: E402
                     bytes_per_sample, have_ref, oracle_encode, pack_samples, random_walk_samples, ref_lib)


def drive(lib, kind, data, params, plan, out_room_total, ample=False):
    """kind 'encode' / 'decode'; plan = [(input bytes offered, output room offered), ...] cycled until the input is
    used up, then flush calls until nothing comes any more.  Returns (last rc, bytes)."""
    st = AecStream()
    st.bits_per_sample, st.block_size, st.rsi, st.flags = params
    init, call, end = (getattr(lib, f"aec_{kind}_{x}") if x else getattr(lib, f"aec_{kind}") for x in ("init", "", "end"))
    for f in (init, end):
        f.restype = C.c_int
        f.argtypes = [C.POINTER(AecStream)]
    call.restype = C.c_int
    call.argtypes = [C.POINTER(AecStream), C.c_int]
    assert init(C.byref(st)) == AEC_OK
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    out = bytearray()
    pos, rc, step, done = 0, AEC_OK, 0, False
    buf = np.zeros(out_room_total if ample else 1 << 20, dtype=np.uint8)       # (ample: everything in ONE call)
    while rc == AEC_OK and len(out) < out_room_total and not done:
        n_in, room = plan[step % len(plan)]
        step += 1
        n_in = min(n_in, src.size - pos)
        room = buf.size if ample else min(room, buf.size, out_room_total - len(out))
        st.next_in = src.ctypes.data + pos
        st.avail_in = n_in
        st.next_out = buf.ctypes.data
        st.avail_out = room
        rc = call(C.byref(st), AEC_FLUSH if pos + n_in >= src.size else AEC_NO_FLUSH)
        used, got = n_in - st.avail_in, room - st.avail_out
        pos += used
        out += buf[:got].tobytes()
        # finished: everything offered, and a call that brought no input produced nothing (the product hands out in
        # batches: include/libaec.h says when)
        done = pos >= src.size and n_in == 0 and got == 0
        if kind == "encode":            # (the flush is complete when its call leaves room unused: encode.c:686-695)
            done = pos >= src.size and st.avail_in == 0 and st.avail_out > 0
    end(C.byref(st))
    return rc, bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="1..4 M samples per case, pieces of 4 KiB .. 3 MiB")
    ap.add_argument("--dump", default="", help="directory: write every case's (damaged) stream there")
    ap.add_argument("--ref-only", action="store_true", help="drive the reference on both sides (case generation without a GPU)")
    return run(ap.parse_args())


def run(args):
    if not have_ref():
        print("oracle/_ref not built: nothing to compare with")
        return 0
    ref = ref_lib()
    if args.ref_only:
        prod = ref
    else:
        import torch  # noqa: F401
        from libaec_amd import api
        prod = api.library()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        bps = int(rng.choice([8, 16, 16, 32]))
        bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 16, 128, 512]))
        flags = AEC_DATA_PREPROCESS if rng.random() < 0.85 else 0
        if rng.random() < 0.3:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.3:
            flags |= AEC_DATA_SIGNED
        nb = bytes_per_sample(bps, flags)
        n = int(rng.choice([1000000, 2500000, 4000000] if args.big else [3000, 30000, 120000]))
        vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.5, 3.0, 50.0])), zero_frac=0.2, jump_frac=0.002)
        data = pack_samples(vals, bps, flags)
        params = (bps, bs, rsi, flags)
        ch = [4096, 70000, 300000, 1 << 20, 3 << 20] if args.big else [nb * 7, 64, 1000, 4096, 70000, 1 << 20]
        plan = [(int(rng.choice(ch)), int(rng.choice(ch))) for _ in range(int(rng.integers(1, 6)))]
        why = ""
        # encode: same calls, same stream
        cap = data.size * 2 + 4096
        rc_r, enc_r = drive(ref, "encode", data, params, plan, cap)
        rc_p, enc_p = drive(prod, "encode", data, params, plan, cap)
        # (a flush call BEHIND the one that completed the stream makes the reference write its last byte again,
        # encode.c:686-695 -- which the driver above does when the reference's output happened to end exactly on the
        # end of a buffer.  The product hands out in batches and never repeats the byte: the stream proper is compared.)
        if rc_r == rc_p and enc_r == enc_p + enc_p[-1:] and enc_p:
            print(f"case {case}: the reference repeated its last byte on a flush call behind the end", flush=True)
            enc_r = enc_p
        if (rc_r, enc_r) != (rc_p, enc_p):
            why = f"encode: reference rc {rc_r} {len(enc_r)} bytes, product rc {rc_p} {len(enc_p)} bytes"
        else:
            enc = bytearray(enc_r)
            damage = int(rng.integers(0, 4))               # 0 none
            if damage == 1:
                for _ in range(int(rng.integers(1, 4))):
                    enc[int(rng.integers(0, len(enc)))] ^= 1 << int(rng.integers(0, 8))
            elif damage == 2:
                enc = enc[: int(rng.integers(1, len(enc)))]
            elif damage == 3:
                o = int(rng.integers(0, len(enc)))
                ln = int(rng.integers(1, 100))
                enc[o:o + ln] = bytes(rng.integers(0, 256, min(ln, len(enc) - o), dtype=np.uint8).tolist())
            out_total = ((n + bs - 1) // bs) * bs * nb
            if args.dump:
                os.makedirs(args.dump, exist_ok=True)
                with open(os.path.join(args.dump, f"s{args.seed}_c{case}_{bps}_{bs}_{rsi}_{flags}_{n}_d{damage}.aec"), "wb") as f:
                    f.write(bytes(enc))
            rc_r, dec_r = drive(ref, "decode", enc, params, plan, out_total)
            rc_p, dec_p = drive(prod, "decode", enc, params, plan, out_total)
            # (on AEC_DATA_ERROR the reference returns without writing out the samples of the RSI it was in, although
            # avail_out has been counted down for them -- decode.c:818-825: only the return code is compared then)
            if rc_r != rc_p or (rc_r == AEC_OK and dec_r != dec_p):
                k = next((i for i in range(min(len(dec_r), len(dec_p))) if dec_r[i] != dec_p[i]), -1)
                # the reference's own answer depends on the room it is given: a zero run that overruns its RSI is refused
                # on the fast path only (decode.c:543-544; m_zero_output has no such check), i.e. when avail_out holds the
                # whole run.  The arbiter is therefore the reference in ONE call with ample room, which the product must
                # match both in pieces and in one call.
                one = [(1 << 30, 1 << 20)]
                rc_r1, dec_r1 = drive(ref, "decode", enc, params, one, out_total + (1 << 20), ample=True)
                rc_p1, dec_p1 = drive(prod, "decode", enc, params, one, out_total)
                k1 = next((i for i in range(min(len(dec_r1), len(dec_p1))) if dec_r1[i] != dec_p1[i]), -1)
                if rc_r != rc_p and rc_r1 == rc_p and rc_p1 == rc_p:
                    print(f"case {case}: reference rc {rc_r} with this room, {rc_r1} in one call with ample room; product {rc_p} both ways",
                          flush=True)
                    continue
                why = (f"decode (damage {damage}): reference rc {rc_r} {len(dec_r)} bytes, product rc {rc_p} {len(dec_p)} bytes, "
                       f"first difference at {k}; in ONE call: reference rc {rc_r1} {len(dec_r1)}, product rc {rc_p1} {len(dec_p1)}, "
                       f"first difference at {k1}; RSI bytes {rsi * bs * nb}")
        print(f"case {case}: bps {bps} bs {bs} rsi {rsi} flags {flags} n {n} plan {plan}: {'ok' if not why else 'MISMATCH ' + why}",
              flush=True)
        bad += 1 if why else 0
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
