"""Ad-hoc GPU diagnostics (not a test): locate the first divergence between the HIP encoder and
the oracle for one input."""
import sys
import numpy as np
import torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import *
from libaec_amd import gpu


def diagnose(name, data, bps, bs, rsi, flags):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    rc, want, trace, offs, bits = oracle_encode(data, bps, bs, rsi, flags, want_trace=True)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data.copy()).cuda()
    d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
    got = d_out[:nbytes].cpu().numpy().tobytes()
    goff = d_off.cpu().numpy().astype(np.uint64)
    print(name, 'bits', tb, bits, 'bytes', nbytes, len(want), 'equal', got == want)
    if got == want:
        return
    bad = np.nonzero(goff[:-1] != offs)[0]
    print(' first RSI with wrong start offset:', bad[:5], 'of', len(offs))
    a = np.frombuffer(got, np.uint8); b = np.frombuffer(want, np.uint8)
    m = min(len(a), len(b))
    diff = np.nonzero(a[:m] != b[:m])[0]
    if len(diff):
        fb = int(diff[0]); print(' first differing byte', fb, 'bit', fb * 8, hex(a[fb]), hex(b[fb]))
        r = int(np.searchsorted(offs, fb * 8, side='right') - 1)
        print(' in RSI', r, 'rsi start bit', offs[r])
        # which block?  accumulate oracle trace bits
        nblk_rsi = rsi
        start = r * rsi
        acc = int(offs[r])
        for j in range(start, min(start + rsi, len(trace))):
            nb_ = int(trace['bits'][j])
            if acc + nb_ > fb * 8:
                print(' block', j, 'in-rsi', j - start, 'opt', trace['option'][j], 'k', trace['k'][j], 'bits', nb_, 'block start bit', acc)
                print(' neighbours opts', trace['option'][max(start, j - 3):j + 4], 'k', trace['k'][max(start, j - 3):j + 4])
                break
            acc += nb_


if __name__ == '__main__':
    rng = np.random.default_rng(1)
    n = 16 * 128 * 37 + 5
    x = np.clip(32768 + np.cumsum(rng.integers(-3, 4, size=n)), 0, 65535).astype('<u2')
    x[5000:9000] = x[5000]
    diagnose('smoke', x.view(np.uint8), 16, 16, 128, 8)
    diagnose('smoke-noconst', np.clip(32768 + np.cumsum(rng.integers(-3, 4, size=n)), 0, 65535).astype('<u2').view(np.uint8), 16, 16, 128, 8)
    diagnose('smoke-full', x[:16 * 128 * 37].copy().view(np.uint8), 16, 16, 128, 8)
    diagnose('smoke-3rsi', x[:16 * 128 * 3].copy().view(np.uint8), 16, 16, 128, 8)
    from libaec_amd import api
    data = x.view(np.uint8)
    rc, want, trace, offs, bits = oracle_encode(data, 16, 16, 128, 8, want_trace=True)
    rc, got = api.aec_buffer_encode(data, 16, 16, 128, 8)
    print('abi', rc, len(got), len(want), got == want)
    a = np.frombuffer(got, np.uint8); b = np.frombuffer(want, np.uint8)
    m = min(len(a), len(b)); diff = np.nonzero(a[:m] != b[:m])[0]
    print('first diffs', diff[:10], 'last RSI start bit', offs[-1], 'total bits', bits, 'trace tail', trace[-3:])
    if len(diff): print([hex(v) for v in a[diff[0]-2:diff[0]+6]], [hex(v) for v in b[diff[0]-2:diff[0]+6]])
