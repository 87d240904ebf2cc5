"""SZIP entry points (libsz.so.2, BASELINE config 5) on the GPU against vectors the reference's
own shim produced (tests/golden/sz_vectors.npz) and, when oracle/_ref travelled, the reference
shim itself on larger HDF5-style chunks."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from helpers import REF_SO, have_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def szip():
    import torch
    assert torch.cuda.is_available()
    from libaec_amd import szip as s
    s.library()
    return s


def cases():
    z = np.load(os.path.join(GOLDEN_DIR, "sz_vectors.npz"))
    for i, name in enumerate(z["names"]):
        opts, bpp, ppb, pps = (int(v) for v in z["params"][i])
        data = z["inputs"][int(z["in_off"][i]):int(z["in_off"][i + 1])]
        comp = z["outputs"][int(z["out_off"][i]):int(z["out_off"][i + 1])].tobytes()
        yield str(name), opts, bpp, ppb, pps, data, comp


def test_sz_golden_vectors(szip):
    assert szip.library().SZ_encoder_enabled() == 1
    for name, opts, bpp, ppb, pps, data, comp in cases():
        rc, got = szip.compress(data, data.size * 2 + 4096, opts, bpp, ppb, pps)
        assert rc == szip.SZ_OK and got == comp, name
        rc, dec = szip.decompress(comp, data.size, opts, bpp, ppb, pps)
        assert rc == szip.SZ_OK and dec == data.tobytes(), name
    # output buffer too small -> SZ_OUTBUFF_FULL (reference sz_compat.c:171-172)
    name, opts, bpp, ppb, pps, data, comp = next(cases())
    rc, got = szip.compress(data, 64, opts, bpp, ppb, pps)
    assert rc == szip.SZ_OUTBUFF_FULL and got == comp[:64]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_sz_hdf5_style_chunks_vs_reference_shim(szip):
    """config 5 shape: 1 MiB chunks of 8-bit pixels, 8 px/block, 1024 px/scanline, NN (+MSB)."""
    from test_gpu_parity import gen
    ref = szip.bind(C.CDLL(REF_SO))
    data = gen(2, 4 << 20)
    for opts in (szip.SZ_NN_OPTION_MASK | szip.SZ_RAW_OPTION_MASK,
                 szip.SZ_NN_OPTION_MASK | szip.SZ_MSB_OPTION_MASK | szip.SZ_RAW_OPTION_MASK):
        for c in range(4):
            chunk = data[c << 20:(c + 1) << 20]
            rc_r, want = szip.compress(chunk, chunk.size * 2, opts, 8, 8, 1024, lib=ref)
            rc, got = szip.compress(chunk, chunk.size * 2, opts, 8, 8, 1024)
            assert (rc, got) == (rc_r, want)
            rc, dec = szip.decompress(got, chunk.size, opts, 8, 8, 1024)
            assert rc == 0 and dec == chunk.tobytes()
    assert len(want) < chunk.size // 2
