"""debug helper (GPU box): streaming decode in pieces, reports the first mismatch and the call history"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import *
from libaec_amd.api import Decoder
PP, MSB, SGN = AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED
rng = np.random.default_rng(int(os.environ.get("AEC_SWEEP_SEED", "17")))
for bps, bs, rsi, flags in ((16, 16, 128, PP), (8, 8, 128, PP | MSB), (12, 32, 64, PP | SGN)):
    nb = bytes_per_sample(bps, flags)
    n = bs * rsi * 37 + 5 * bs + 3
    vals = random_walk_samples(rng, n, bps, flags, scale=4.0, zero_frac=0.15)
    data = pack_samples(vals, bps, flags)
    rc, whole, *_ = oracle_encode(data, bps, bs, rsi, flags)
    nblk = (n + bs - 1) // bs
    rc, want, _ = oracle_decode(whole, bps, bs, rsi, flags, nblk * bs * nb)
    d = Decoder(bps, bs, rsi, flags)
    dec = bytearray()
    pos = 0
    hist = []
    while pos < len(whole):
        step = int(rng.choice([1, 700, 3000, 20000, 60000]))
        chunk = whole[pos:pos + step]
        off = 0
        while True:
            room = min(int(rng.choice([nb, 1000 * nb, 1 << 20])), len(want) - len(dec))
            rc, used, got = d.call(chunk[off:], room, AEC_NO_FLUSH)
            hist.append((pos + off, len(chunk) - off, room, rc, used, len(got), len(dec)))
            assert rc == AEC_OK
            dec += got
            off += used
            if off >= len(chunk) and not got:
                break
        pos += step
    d.end()
    ok = bytes(dec) == want
    print((bps, bs, rsi, flags), "OK" if ok else "MISMATCH", len(dec), len(want), "rsi_bytes", bs * rsi * nb)
    if not ok:
        m = next((i for i in range(min(len(dec), len(want))) if dec[i] != want[i]), None)
        print("first mismatch at byte", m, "sample", None if m is None else m // nb, "block", None if m is None else m // nb // bs)
        for h in hist:
            if m is None or h[6] + h[5] >= m - 200000:
                print("  in_pos %d in_len %d room %d rc %d used %d got %d dec_before %d" % h)
                if m is not None and h[6] > m + 100000:
                    break
