"""Edge cases of the single-pass encoder (k_encode_fused), run on the GPU box by
tests/test_gpu_parity.py::test_fused_encoder_edges under several AEC_FUSED_SEGS / AEC_FUSED_PARTS
settings with AEC_ENC_FUSED=1 (and once without): streams whose waves / partitions begin and end inside
one 32-bit word (long stretches of zero blocks), partitions of every fill, ragged ends, every
container size -- against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import (AEC_DATA_3BYTE, AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, AEC_OK,  # noqa: E402
                     bytes_per_sample, oracle_encode, pack_samples)
from libaec_amd import api  # noqa: E402

PP, MSB, SGN = AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED


def main():
    rng = np.random.default_rng(int(os.environ.get("AEC_SWEEP_SEED", "4242")))
    n_cases = 0
    for bps, bs, rsi, flags in ((8, 8, 64, 0), (8, 8, 128, PP), (16, 16, 128, PP), (16, 16, 64, 0), (32, 64, 64, 0),
                                (32, 32, 4096, PP | MSB | SGN), (24, 64, 100, PP | AEC_DATA_3BYTE), (12, 32, 7, PP),
                                (32, 64, 1, 0), (16, 64, 2, PP | MSB)):
        nb = bytes_per_sample(bps, flags)
        seg = 64 * bs
        for kind in ("zeros", "const", "mixed", "noise_islands"):
            for n in (1, seg - 1, seg, 3 * seg + 5, 16 * seg, 16 * seg + 1, 37 * seg + bs + 3, 260 * seg, 1031 * seg + 17):
                if n * nb > (24 << 20):
                    continue
                if kind == "zeros":
                    vals = np.zeros(n, dtype=np.int64)
                elif kind == "const":
                    vals = np.full(n, (1 << (bps - 1)) - 3, dtype=np.int64)
                elif kind == "mixed":
                    vals = np.zeros(n, dtype=np.int64)
                    # stretches of zero blocks of every length between short busy stretches
                    pos = 0
                    while pos < n:
                        z = int(rng.integers(1, 40 * seg))
                        pos += z
                        b = int(rng.integers(1, 3 * bs))
                        vals[pos:pos + b] = rng.integers(0, 1 << min(bps - 1, 9), size=len(vals[pos:pos + b]))
                        pos += b
                else:
                    vals = np.zeros(n, dtype=np.int64)
                    for _ in range(max(1, n // (50 * seg))):
                        at = int(rng.integers(0, n))
                        ln = int(rng.integers(1, 2 * seg))
                        vals[at:at + ln] = rng.integers(0, 1 << (bps - 1), size=len(vals[at:at + ln]))
                data = pack_samples(vals, bps, flags)
                rc_o, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
                rc, got = api.aec_buffer_encode(data, bps, bs, rsi, flags)
                assert rc == rc_o == AEC_OK, (bps, bs, rsi, flags, kind, n, rc)
                if got != want:
                    m = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
                    raise AssertionError((bps, bs, rsi, flags, kind, n, len(got), len(want), m))
                n_cases += 1
    print("fused edges ok:", n_cases, "cases; AEC_ENC_FUSED=%s AEC_FUSED_SEGS=%s AEC_FUSED_PARTS=%s" % (
        os.environ.get("AEC_ENC_FUSED"), os.environ.get("AEC_FUSED_SEGS"), os.environ.get("AEC_FUSED_PARTS")))


if __name__ == "__main__":
    main()
