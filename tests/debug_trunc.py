"""debug helper (GPU box): time every decode of the truncated-stream test, print the slowest"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import *
from libaec_amd import api
PP, MSB, SGN = AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED
rng = np.random.default_rng(2718)
rows = []
for it in range(40):
    bps = int(rng.choice([8, 12, 16, 24, 32]))
    flags = PP if rng.random() < 0.8 else 0
    if rng.random() < 0.5:
        flags |= MSB
    if rng.random() < 0.4:
        flags |= SGN
    bs = int(rng.choice([8, 16, 32, 64]))
    rsi = int(rng.choice([1, 3, 16, 64, 130]))
    nb = bytes_per_sample(bps, flags)
    n = int(rng.integers(bs, 6000))
    vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 3, 50, 5000])),
                               zero_frac=float(rng.choice([0.05, 0.5])))
    data = pack_samples(vals, bps, flags)
    rc, enc, *_ = oracle_encode(data, bps, bs, rsi, flags)
    cap = ((n + bs - 1) // bs) * bs * nb
    for cut in [int(v) for v in rng.integers(0, len(enc) + 1, 4)]:
        t0 = time.perf_counter()
        rc, dec = api.aec_buffer_decode(enc[:cut], bps, bs, rsi, flags, cap)
        dt = time.perf_counter() - t0
        rows.append((dt, it, bps, bs, rsi, flags, n, len(enc), cut, len(dec)))
        first = int(rng.integers(0, cut + 1))
        from libaec_amd.api import Decoder
        d = Decoder(bps, bs, rsi, flags)
        ncall = 0
        t0 = time.perf_counter()
        worst = 0
        for piece in (enc[:first], enc[first:cut], b""):
            off = 0
            while True:
                t1 = time.perf_counter()
                rc, used, out = d.call(piece[off:], cap, AEC_NO_FLUSH if piece else AEC_FLUSH)
                worst = max(worst, time.perf_counter() - t1)
                ncall += 1
                off += used
                if off >= len(piece) and not out:
                    break
        d.end()
        rows.append((time.perf_counter() - t0, -it, bps, bs, rsi, flags, n, ncall, first, int(worst * 1e6)))
rows.sort(reverse=True)
print("total", sum(r[0] for r in rows))
for r in rows[:15]:
    print("%.4f s it %d bps %d bs %d rsi %d flags %d n %d enc %d cut %d dec %d" % r)
