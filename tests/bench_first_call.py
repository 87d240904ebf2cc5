#!/usr/bin/env python3
"""The first calls of a process (GPU box): tests/bench_first_call.py [torch]
Without an argument the first aec_buffer_encode also initialises the HIP runtime (hipInit: 90 .. 150 ms on the test box, any
HIP program's); with `torch` the runtime is up before the library is loaded, and the first encode / first decode show what
is the library's own: its code objects (all instantiations of a translation unit load with the first kernel of it -- a
second parameter set costs nothing more) and the context's workspace."""
import ctypes as C, time, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
t3 = time.perf_counter()
if len(sys.argv) > 1:
    import torch
    torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
    print(f'torch cuda init {1e3*(time.perf_counter()-t3):.1f} ms')
    t3 = time.perf_counter()
from libaec_amd import api
t4 = time.perf_counter()
lib = api.library()
t5 = time.perf_counter()
print(f"import api {1e3*(t4-t3):.1f} ms, dlopen libaec {1e3*(t5-t4):.1f} ms")
rng = np.random.default_rng(1)
data = (np.cumsum(rng.integers(-3, 4, 32768)) + 1000).astype('<u2').view(np.uint8)
for i in range(3):
    t = time.perf_counter(); rc, enc = api.aec_buffer_encode(data, 16, 16, 128, 8); te = time.perf_counter() - t
    t = time.perf_counter(); rc2, dec = api.aec_buffer_decode(enc, 16, 16, 128, 8, data.size); td = time.perf_counter() - t
    print('rc', rc, rc2, len(enc), len(dec), dec == data.tobytes(), (np.frombuffer(dec, dtype=np.uint8) != data).nonzero()[0][:5] if len(dec) == data.size else None)
    print(f"call {i}: encode 64 KiB {1e3*te:.2f} ms, decode {1e3*td:.2f} ms")
data8 = (np.cumsum(rng.integers(-2, 3, 65536)) % 256).astype(np.uint8)
for i in range(2):
    t = time.perf_counter(); rc, enc = api.aec_buffer_encode(data8, 8, 8, 128, 8); te = time.perf_counter() - t
    t = time.perf_counter(); rc2, dec = api.aec_buffer_decode(enc, 8, 8, 128, 8, data8.size); td = time.perf_counter() - t
    print(f"8-bit call {i}: encode 64 KiB {1e3*te:.2f} ms, decode {1e3*td:.2f} ms")
