for m in 0 1 2 4 3 7; do echo -n "dbg=$m "; AEC_DBG_PACK=$m python - <<'PY'
import sys, os, numpy as np, ctypes as C, torch
sys.path.insert(0,'.')
from libaec_amd import gpu
lib = C.CDLL('libaec_amd/lib/libaec_datagen.so')
n = 1<<30
a = np.empty(n, dtype=np.uint8)
lib.aec_gen_fill_parallel(C.c_uint(0), C.c_uint64(0), C.c_void_p(a.ctypes.data), C.c_size_t(n//2), C.c_uint(16))
codec = gpu.Codec(16,16,128,8); codec.reserve(n)
d_in = torch.from_numpy(a).cuda()
d_out = torch.empty(codec.encode_bound(n), dtype=torch.uint8, device='cuda')
d_off = torch.empty(codec.rsi_count(n)+1, dtype=torch.int64, device='cuda')
d_res = torch.zeros(16, dtype=torch.uint8, device='cuda')
l = gpu._lib(); l.aec_gpu_profile.argtypes=[C.c_void_p,C.c_int]; l.aec_gpu_phase_ms.argtypes=[C.c_void_p,C.POINTER(C.c_float)]
l.aec_gpu_profile(codec.ctx,1)
acc = np.zeros(5)
for i in range(6):
    codec.encode_async(d_in, n, d_out, d_off, d_res)
    ms=(C.c_float*5)(); l.aec_gpu_phase_ms(codec.ctx, ms)
    if i: acc += np.array(list(ms))
print('analyze %.3f pack %.3f' % (acc[0]/5, acc[3]/5))
PY
done
