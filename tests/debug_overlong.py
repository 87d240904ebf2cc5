#!/usr/bin/env python3
"""Diagnostics (GPU box): index and decode records for a crafted stream with over-long coded data sets."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import oracle_decode, oracle_encode, bytes_per_sample, AEC_DATA_PREPROCESS as PP
from libaec_amd import gpu
from helpers import craft_overlong_stream
bps, bs, rsi, n_rsi, long_hi = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (16, 16, 8, 200, 120)
rng = np.random.default_rng(bps + bs)
enc = craft_overlong_stream(rng, bps, bs, rsi, n_rsi, {8: 3, 16: 4, 32: 5}[bps], 0.05, long_hi)
nbytes = n_rsi * rsi * bs * bytes_per_sample(bps, PP)
rc_o, dec_o, _ = oracle_decode(enc, bps, bs, rsi, PP, nbytes)
dev = torch.device("cuda", 0)
codec = gpu.Codec(bps, bs, rsi, PP)
d_in = torch.from_numpy(np.frombuffer(enc + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
d_off = torch.zeros(n_rsi + 2, dtype=torch.int64, device=dev)
d_res = torch.zeros(40, dtype=torch.uint8, device=dev)
import ctypes as C
lib = gpu._lib()
if os.environ.get("HINT"):
    lib.aec_gpu_set_index_hint.argtypes = [C.c_void_p, C.c_uint64]
    lib.aec_gpu_set_index_hint(codec.ctx, C.c_uint64(int(os.environ["HINT"])))
codec.index_async(d_in, len(enc), 0, d_off, n_rsi, d_res)
torch.cuda.synchronize()
print("index record", np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=gpu.DEC_RESULT_DTYPE if hasattr(gpu, "DEC_RESULT_DTYPE") else np.uint32))
off = d_off.cpu().numpy()
print("offsets", off[:6], "...")
d_out = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=dev)
d_res2 = torch.zeros(40, dtype=torch.uint8, device=dev)
codec.decode_async(d_in, len(enc), d_off, n_rsi, n_rsi * rsi, d_out, d_res2)
torch.cuda.synchronize()
print("decode record", np.frombuffer(d_res2.cpu().numpy().tobytes(), dtype=np.uint32))
got = d_out.cpu().numpy()[:nbytes].tobytes()
bad = [i for i in range(0, nbytes, rsi * bs * 2) if got[i:i + rsi * bs * 2] != dec_o[i:i + rsi * bs * 2]]
print("wrong RSIs:", len(bad), bad[:5])
