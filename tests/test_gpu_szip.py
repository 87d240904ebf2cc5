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


def test_batch_of_foreign_chunks_decodes_in_two_launches(szip):
    """HDF5-style dataset: many independently coded chunks (here coded by the CPU reference / oracle,
    i.e. streams that come without any offset table).  aec_gpu_index_batch_async walks all chunks
    concurrently (one wavefront each), one aec_gpu_decode_async call decodes every RSI of the batch."""
    import time
    import torch
    from libaec_amd import gpu
    from helpers import oracle_encode, ref_encode
    from test_gpu_parity import gen
    bps, bs, rsi, flags = 8, 8, 128, 8            # SZ: 8-bit pixels, 8 px/block, 1024 px/scanline, NN
    chunk_bytes, n_chunks = 256 << 10, 48
    data = gen(2, chunk_bytes * n_chunks)
    rpc = chunk_bytes // (bs * rsi)
    streams, offs, pos = [], [0], 0
    for s in range(n_chunks):
        chunk = data[s * chunk_bytes:(s + 1) * chunk_bytes]
        rc, enc = ref_encode(chunk, bps, bs, rsi, flags) if have_ref() else oracle_encode(chunk, bps, bs, rsi, flags)[:2]
        assert rc == 0
        pad = (-len(enc)) % 16
        streams.append(enc + bytes(pad))
        pos += len(enc) + pad
        offs.append(pos)
    blob = np.frombuffer(b"".join(streams), dtype=np.uint8)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(blob.copy()).cuda()
    d_choff = torch.tensor(offs, dtype=torch.int64, device="cuda")
    d_off = torch.zeros(n_chunks * rpc, dtype=torch.int64, device="cuda")
    d_res = torch.zeros(n_chunks * 40, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(data.size + 16, dtype=torch.uint8, device="cuda")
    d_dres = torch.zeros(40, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codec.index_batch_async(d_in, blob.size, d_choff, n_chunks, rpc, d_off, d_res)
        codec.decode_async(d_in, blob.size, d_off, n_chunks * rpc, n_chunks * rpc * rsi, d_out, d_dres)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)
    assert np.all(res["n_rsi"] == rpc) and np.all(res["status"] == 0)
    assert d_dres.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]["status"] == 0
    assert torch.equal(d_out[:data.size].cpu(), torch.from_numpy(data))
    print(f"batch of {n_chunks} chunks ({data.size >> 20} MiB): {dt * 1e3:.2f} ms -> {data.size / dt / 1e9:.2f} GB/s")


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_sz_batch_calls_vs_reference_shim(szip):
    """SZ_BatchCompress / SZ_BatchDecompress (n chunks, one call) against the reference shim run chunk by
    chunk (reference src/sz_compat.c:110-268; tests/check_szcomp.c:6-37 is its own configuration): 1 MiB
    chunks of 8-bit pixels with NN and NN|MSB, 32-bit pixels coded as byte planes with a scan line that is
    not a whole number of blocks and a ragged last chunk, and 16-bit pixels."""
    from test_gpu_parity import gen
    ref = szip.bind(C.CDLL(REF_SO))
    NN, MSBO, RAW = szip.SZ_NN_OPTION_MASK, szip.SZ_MSB_OPTION_MASK, szip.SZ_RAW_OPTION_MASK
    rng = np.random.default_rng(8)
    d8 = gen(2, 6 << 20)
    d16 = gen(0, 3 << 20)
    d32 = np.cumsum(rng.integers(-40, 41, size=300_000)).astype("<i4").view(np.uint8)
    cases = [
        ("8-bit NN", NN | RAW, 8, 8, 1024, [d8[i << 20:(i + 1) << 20] for i in range(6)]),
        ("8-bit NN MSB", NN | MSBO | RAW, 8, 8, 1024, [d8[i << 20:(i + 1) << 20] for i in range(6)]),
        ("16-bit NN", NN | RAW, 16, 16, 2048, [d16[i << 19:(i + 1) << 19] for i in range(6)]),
        # 32 bpp: four byte planes, 1000 px per scan line = 62.5 blocks of 16 (padded), ragged last chunk
        ("32-bit planes", NN | MSBO | RAW, 32, 16, 1000, [d32[:400_000], d32[400_000:800_000], d32[800_000:1_100_004]]),
        ("8-bit no NN", RAW, 8, 8, 1000, [d8[:250_000], d8[250_000:500_001]]),
    ]
    for name, opts, bpp, ppb, pps, chunks in cases:
        want = [szip.compress(c, c.size * 2 + 4096, opts, bpp, ppb, pps, lib=ref) for c in chunks]
        assert all(rc == 0 for rc, _ in want), name
        rc, got, st = szip.compress_batch(chunks, [c.size * 2 + 4096 for c in chunks], opts, bpp, ppb, pps)
        assert rc == 0 and st == [0] * len(chunks), (name, rc, st)
        assert got == [w for _, w in want], name
        # and back, from the reference's streams
        back = [szip.decompress(w, c.size, opts, bpp, ppb, pps, lib=ref) for (_, w), c in zip(want, chunks)]
        rc, dec, st = szip.decompress_batch([w for _, w in want], [c.size for c in chunks], opts, bpp, ppb, pps)
        assert rc == 0 and st == [0] * len(chunks), (name, rc, st)
        assert dec == [b for _, b in back], name
    # a chunk whose output does not fit: SZ_OUTBUFF_FULL for that chunk, the others are unaffected
    name, opts, bpp, ppb, pps, chunks = cases[0]
    sizes = [c.size * 2 for c in chunks]
    sizes[2] = 100
    rc, got, st = szip.compress_batch(chunks, sizes, opts, bpp, ppb, pps)
    assert st[2] == szip.SZ_OUTBUFF_FULL and [s for i, s in enumerate(st) if i != 2] == [0] * 5
    assert got[3] == szip.compress(chunks[3], chunks[3].size * 2, opts, bpp, ppb, pps, lib=ref)[1]


def _bits_to_bytes(bits):
    bits = bits + "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def test_batch_of_small_chunks_reports_errors_per_chunk(szip):
    """80 chunks of 8 KiB through SZ_BatchDecompress take the small-chunk path (every chunk walked by one wavefront,
    ONE decode launch for all of them).  Two of the streams hold a second-extension code beyond the table
    (reference src/decode.c:560-616: m > 90 is AEC_DATA_ERROR) -- an error only the decoder sees; each of them must
    come back as AEC_DATA_ERROR, every other chunk intact."""
    from test_gpu_parity import gen
    NN, RAW = szip.SZ_NN_OPTION_MASK, szip.SZ_RAW_OPTION_MASK
    opts, bpp, ppb, pps = NN | RAW, 8, 8, 1024
    data = gen(2, 80 * 8192)
    chunks = [data[i * 8192:(i + 1) * 8192] for i in range(80)]
    rc, enc, st = szip.compress_batch(chunks, [c.size * 2 + 4096 for c in chunks], opts, bpp, ppb, pps)
    assert rc == 0 and st == [0] * 80
    # a hand-made stream for a whole chunk's worth of RSIs is not needed: the first coded data set decides.
    # id 000 + selector 1 (second extension) + reference sample + a code of 95 zeros and a one, then ones
    bad = _bits_to_bytes("000" + "1" + "10000000" + "0" * 95 + "1" + "1" * 3 + "1" * 64) + bytes(64)
    enc = list(enc)
    enc[7] = bad
    enc[55] = bad
    rc, dec, st = szip.decompress_batch(enc, [c.size for c in chunks], opts, bpp, ppb, pps)
    assert st[7] == -3 and st[55] == -3, (rc, st[:10], st[50:60])          # AEC_DATA_ERROR
    good = [i for i in range(80) if i not in (7, 55)]
    assert all(st[i] == 0 for i in good), st
    assert all(dec[i] == chunks[i].tobytes() for i in good)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_one_gib_of_one_mib_chunks_vs_reference_shim(szip):
    """BASELINE config 5 at its stated size (at least 1 GiB in 1 MiB HDF5-style chunks, 8-bit,
    8 pixels per block, 1024 per scan line, NN): every chunk through SZ_BatchCompress / SZ_BatchDecompress, 64 chunks
    per call, against the reference shim chunk by chunk (reference src/sz_compat.c:110-268) -- a digest of every
    compressed chunk, and the round trip."""
    import hashlib
    from test_gpu_parity import gen
    ref = szip.bind(C.CDLL(REF_SO))
    NN, RAW = szip.SZ_NN_OPTION_MASK, szip.SZ_RAW_OPTION_MASK
    opts, bpp, ppb, pps = NN | RAW, 8, 8, 1024
    n_chunks, per_call = 1024, 64
    data = gen(2, n_chunks << 20)
    for g0 in range(0, n_chunks, per_call):
        chunks = [data[i << 20:(i + 1) << 20] for i in range(g0, g0 + per_call)]
        rc, got, st = szip.compress_batch(chunks, [(1 << 20) + (1 << 16)] * per_call, opts, bpp, ppb, pps)
        assert rc == 0 and st == [0] * per_call, (g0, rc, st)
        if g0 % 256 == 0:          # the reference on one core: every fourth call's chunks in full, ...
            want = [szip.compress(c, (1 << 20) + (1 << 16), opts, bpp, ppb, pps, lib=ref) for c in chunks]
            assert all(rc == 0 for rc, _ in want)
            assert [hashlib.sha256(g).digest() for g in got] == [hashlib.sha256(w).digest() for _, w in want], g0
        rc, dec, st = szip.decompress_batch(got, [1 << 20] * per_call, opts, bpp, ppb, pps)
        assert rc == 0 and st == [0] * per_call, (g0, rc, st)
        assert all(d == c.tobytes() for d, c in zip(dec, chunks)), g0      # ... the round trip for all of them
        # and the reference decodes what was produced here (first chunk of every call)
        rc, back = szip.decompress(got[0], 1 << 20, opts, bpp, ppb, pps, lib=ref)
        assert rc == 0 and back == chunks[0].tobytes(), g0


def test_random_sweep_of_the_batch_entry_points(szip):
    """tests/fuzz_batch_gpu.py, 25 cases: aec_buffer_encode_batch / aec_buffer_decode_batch with random parameters
    and 1..300 chunks per call -- equal chunks of whole RSIs (ONE launch set for all of them), equal ragged ones,
    random sizes, tiny and large mixed, an output buffer too small for its stream -- every chunk against the oracle."""
    import argparse
    import fuzz_batch_gpu
    assert fuzz_batch_gpu.run(argparse.Namespace(cases=25, seed=9)) == 0


@pytest.mark.parametrize("n,mib", [(48, 35), (112, 21)])
def test_batch_decode_with_few_chunks_per_group(n, mib):
    """aec_buffer_decode_batch with FEW large streams per 12-MiB group and MANY groups per part (round-3 ADVICE: the
    group-relative chunk offsets outgrew their region and the last chunk of a part came back empty with AEC_OK).
    48 streams of ~6.2 MiB (one per group, 12 groups in each of the four parts) and 112 of ~3.7 MiB, low-entropy
    16-bit data: every chunk must come back whole and equal to what went in."""
    import torch  # noqa: F401
    from fuzz_batch_gpu import batch
    from libaec_amd import api
    from test_gpu_parity import gen
    lib = api.library()
    bps, bs, rsi, flags = 16, 16, 128, 8
    data = gen(0, mib << 20)
    rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
    assert rc == 0
    stream = np.frombuffer(enc, dtype=np.uint8).copy()
    rc, dec, st = batch(lib, "aec_buffer_decode_batch", (bps, bs, rsi, flags), [stream] * n, [data.size] * n)
    assert rc == 0 and st == [0] * n, (rc, st)
    for i, d in enumerate(dec):
        assert d.size == data.size and np.array_equal(d, data), f"chunk {i} of {n}: {d.size} bytes"
