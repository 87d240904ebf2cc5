"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI
(libaec.so.0 via ctypes), against the oracle, the golden vectors the reference produced and --
at full size -- through round trips and hashes."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from helpers import (AEC_DATA_3BYTE, AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED,
                     AEC_FLUSH, AEC_MEM_ERROR, AEC_NO_FLUSH, AEC_NOT_ENFORCE, AEC_OK,
                     AEC_RESTRICTED, AEC_STREAM_ERROR, ROOT, bytes_per_sample, have_ref,
                     max_encoded_size, oracle_decode, oracle_encode, pack_samples,
                     random_walk_samples, ref_decode, ref_encode, ref_lib, unpack_samples, craft_overlong_stream)

pytestmark = pytest.mark.gpu

PP, MSB, SGN = AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    from libaec_amd import api as a
    a.library()
    return a


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401
    from libaec_amd import gpu as g
    return g


def gen(kind, nbytes, shard=0):
    lib = C.CDLL(f"{ROOT}/libaec_amd/lib/libaec_datagen.so")
    a = np.empty(nbytes, dtype=np.uint8)
    bps = {0: 2, 1: 4, 2: 1}[kind]
    lib.aec_gen_fill_parallel(C.c_uint(kind), C.c_uint64(shard), C.c_void_p(a.ctypes.data),
                              C.c_size_t(nbytes // bps), C.c_uint(8))
    return a


def check_roundtrip(api, name, bps, bs, rsi, flags, data, expect=None):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if expect is None:
        rc, expect, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK, name
    rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
    assert rc == AEC_OK, (name, rc)
    assert enc == expect, (name, len(enc), len(expect))
    nb = bytes_per_sample(bps, flags)
    nblk = (data.size // nb + bs - 1) // bs
    rc_o, dec_o, _ = oracle_decode(expect, bps, bs, rsi, flags, nblk * bs * nb)
    rc, dec = api.aec_buffer_decode(expect, bps, bs, rsi, flags, nblk * bs * nb)
    assert rc == rc_o == AEC_OK, (name, rc, rc_o)
    assert dec == dec_o, name


def test_typical_rz_config1(api, typical_rz):
    """BASELINE config 1: decode data/typical.rz (j64/r256), re-encode at j16/r128 and at j64/r256."""
    rc, dec = api.aec_buffer_decode(typical_rz, 16, 64, 256, PP | MSB, 1 << 20)
    assert rc == AEC_OK and hashlib.sha256(dec).hexdigest().startswith("e6e1bf684916")
    rc, enc = api.aec_buffer_encode(dec, 16, 64, 256, PP | MSB)
    assert rc == AEC_OK and enc == typical_rz
    rc, enc2 = api.aec_buffer_encode(dec, 16, 16, 128, PP | MSB)
    assert len(enc2) == 740174 and hashlib.sha256(enc2).hexdigest().startswith("60f1f251f7e6")
    rc, dec2 = api.aec_buffer_decode(enc2, 16, 16, 128, PP | MSB, 1 << 20)
    assert dec2 == dec


def test_golden_vectors(api, golden):
    for i in range(len(golden)):
        name, bps, bs, rsi, flags, data, expect, _ = golden.case(i)
        if bps == 1 and flags & SGN:
            continue
        check_roundtrip(api, name, bps, bs, rsi, flags, data, expect)


def test_random_sweep_vs_oracle(api):
    # AEC_SWEEP_ITERS / AEC_SWEEP_SEED in the environment turn this into a soak test
    rng = np.random.default_rng(int(os.environ.get("AEC_SWEEP_SEED", "2025")))
    for it in range(int(os.environ.get("AEC_SWEEP_ITERS", "150"))):
        bps = int(rng.integers(1, 33))
        flags = 0
        if rng.random() < 0.75:
            flags |= PP
        if rng.random() < 0.5:
            flags |= MSB
        if rng.random() < 0.4 and bps > 1:
            flags |= SGN
        if rng.random() < 0.3:
            flags |= AEC_DATA_3BYTE
        if bps <= 4 and rng.random() < 0.5:
            flags |= AEC_RESTRICTED
        if rng.random() < 0.2:
            flags |= AEC_NOT_ENFORCE
            bs = int(rng.integers(1, 33)) * 2
        else:
            bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 128, 130, 300, 4096]))
        n = int(rng.integers(1, 20000)) if rng.random() < 0.9 else int(rng.integers(20000, 400000))
        mode = rng.integers(0, 3)
        if mode == 0:
            vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 1, 5, 50, 1000])),
                                       zero_frac=float(rng.choice([0.05, 0.6])))
        elif mode == 1:
            lo = -(1 << (bps - 1)) if flags & SGN else 0
            hi = (1 << (bps - 1)) - 1 if flags & SGN else (1 << bps) - 1
            vals = rng.integers(lo, hi + 1, size=n)
        else:
            vals = np.repeat(rng.integers(0, 1 << min(bps, 7), size=n // 97 + 1), 97)[:n]
        data = pack_samples(vals, bps, flags)
        check_roundtrip(api, f"it{it}-n{bps}-j{bs}-r{rsi}-f{flags}-len{n}", bps, bs, rsi, flags, data)


def test_small_blocks_direct_feed(api):
    """Blocks of at most 32 bytes take the encoder's direct path (aec_enc.hip Feeder::DIRECT: every lane loads
    its own block, the sample in front of it comes from the neighbour lane).  Blocks of 8 one-byte samples in
    every relation two neighbouring segments can have: same RSI, the second starting an RSI (reference sample
    slot), short segments in between, odd segment counts per wave, end of data inside a segment; unsigned /
    signed / MSB / without preprocessing (which falls back to the rows in LDS)."""
    rng = np.random.default_rng(88)
    for bps in (8, 7, 3):
        for rsi in (64, 128, 65, 130, 192, 1, 4096):
            for flags in (PP, PP | SGN, 0, PP | MSB | SGN):
                nblk = int(rng.choice([64 * 9, 64 * 16 + 1, 64 * 33 + 63, 64 * 40 + 7, rsi * 5 + 64 * 3]))
                n = 8 * nblk - int(rng.integers(0, 8))
                vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.5, 3.0])), zero_frac=0.2)
                data = pack_samples(vals, bps, flags)
                check_roundtrip(api, f"pair-n{bps}-r{rsi}-f{flags}-len{n}", bps, 8, rsi, flags, data)
    # inputs of some size: a wave walks several segments (prefetch of the next one beside the current)
    for rsi, flags in ((128, PP), (64, PP | SGN), (65, PP), (130, PP | MSB), (4096, 0), (192, PP)):
        n = (16 << 20) + 8 * 37 + 3
        vals = random_walk_samples(rng, n, 8, flags, scale=2.0, zero_frac=0.2)
        data = pack_samples(vals, 8, flags)
        check_roundtrip(api, f"pair-big-r{rsi}-f{flags}", 8, 8, rsi, flags, data)


def test_concurrent_streams_share_the_resource_pool(api):
    """Four host threads run one-shot and streaming calls at the same time (ctypes releases the GIL):
    streams are independent objects (SURVEY 8(b) threading contract) and the device-side resources
    they take from / return to the pool of the ABI layer must never be shared by two live streams."""
    import threading
    cases = []
    rng = np.random.default_rng(99)
    for bps, bs, rsi, flags, n in ((16, 16, 128, PP, 300_000), (8, 8, 128, PP | MSB, 500_000),
                                   (32, 32, 64, PP | SGN | MSB, 120_000), (12, 64, 17, PP, 90_000)):
        vals = random_walk_samples(rng, n, bps, flags, scale=3.0, zero_frac=0.2)
        data = pack_samples(vals, bps, flags)
        rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK
        cases.append((bps, bs, rsi, flags, data, want))
    errors = []

    def work(tid):
        try:
            for it in range(12):
                bps, bs, rsi, flags, data, want = cases[(tid + it) % len(cases)]
                rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
                assert rc == AEC_OK and enc == want, ("encode", tid, it)
                rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, flags, len(data))
                assert rc == AEC_OK and dec == bytes(data), ("decode", tid, it)
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_edge_sizes_and_errors(api):
    # empty input -> single zero byte (reference encode.c:686-695)
    rc, enc = api.aec_buffer_encode(b"", 16, 16, 128, PP)
    assert rc == AEC_OK and enc == b"\x00"
    # one sample, 11 samples (padded to a block), trailing odd byte ignored
    for n, extra in ((1, 0), (11, 0), (40, 1)):
        d = np.concatenate([pack_samples(np.arange(n) * 5, 16, PP), np.zeros(extra, np.uint8)])
        check_roundtrip(api, f"tiny{n}", 16, 16, 128, PP, d)
    # output too small: STREAM_ERROR from aec_encode_end, bytes are a prefix (encode.c:944-945)
    big = pack_samples(np.arange(8192) * 7 % 65536, 16, PP)
    rc, full = api.aec_buffer_encode(big, 16, 16, 128, PP)
    rc, part = api.aec_buffer_encode(big, 16, 16, 128, PP, out_size=100)
    assert rc == AEC_STREAM_ERROR and part == full[:100]
    # decoder: 0 < avail_out < bytes_per_sample at exit -> MEM_ERROR (decode.c:821-823)
    rc, _ = api.aec_buffer_decode(full, 16, 16, 128, PP, 8192 * 2 - 1)
    assert rc == AEC_MEM_ERROR
    # smaller output than the stream holds: exactly that many samples, AEC_OK
    rc, dec = api.aec_buffer_decode(full, 16, 16, 128, PP, 1000)
    assert rc == AEC_OK and dec == big[:1000].tobytes()
    # invalid parameters (reference encode.c:777-794, 843-851)
    assert api.aec_buffer_encode(bytes(64), 0, 16, 128, 0)[0] == api.AEC_CONF_ERROR
    assert api.aec_buffer_encode(bytes(64), 8, 12, 128, 0)[0] == api.AEC_CONF_ERROR
    assert api.aec_buffer_encode(bytes(64), 8, 8, 128, AEC_RESTRICTED)[0] == api.AEC_CONF_ERROR


def test_streaming_matches_one_shot(api):
    """Any chunking of input and output gives the same stream (reference tests/check_aec.c:59-200
    drives the state machines with 1-sample / 1-byte calls)."""
    rng = np.random.default_rng(5)
    for bps, bs, rsi, flags in ((16, 16, 8, PP), (8, 8, 3, PP | MSB), (32, 32, 2, PP | SGN | MSB), (24, 16, 5, PP | AEC_DATA_3BYTE)):
        nb = bytes_per_sample(bps, flags)
        n = bs * rsi * 5 + 7
        data = pack_samples(random_walk_samples(rng, n, bps, flags, scale=3.0, zero_frac=0.3), bps, flags)
        rc, whole, *_ = oracle_encode(data, bps, bs, rsi, flags)
        from libaec_amd.api import Decoder, Encoder
        e = Encoder(bps, bs, rsi, flags)
        out = bytearray()
        pos = 0
        while pos < data.size:
            step = int(rng.choice([nb, 2 * nb, 7 * nb, 64 * nb, 1000 * nb]))
            chunk = data[pos:pos + step]
            off = 0
            while off < chunk.size:
                rc, used, got = e.call(chunk[off:], int(rng.choice([1, 3, 50, 4096])), AEC_NO_FLUSH)
                assert rc == AEC_OK
                out += got
                off += used
                if used == 0 and not got:
                    break
            pos += step
        while True:
            rc, used, got = e.call(b"", int(rng.choice([1, 2, 64])), AEC_FLUSH)
            out += got
            if not got:
                break
        assert e.end() == AEC_OK
        assert bytes(out) == whole, (bps, bs, rsi, flags)

        d = Decoder(bps, bs, rsi, flags)
        nblk = (n + bs - 1) // bs
        dec = bytearray()
        pos = 0
        while pos < len(whole):
            step = int(rng.choice([1, 2, 5, 100, 3000]))
            chunk = whole[pos:pos + step]
            off = 0
            while True:
                rc, used, got = d.call(chunk[off:], int(rng.choice([nb, 3 * nb, 1000 * nb])), AEC_NO_FLUSH)
                assert rc == AEC_OK
                dec += got
                off += used
                if off >= len(chunk) and not got:
                    break
            pos += step
        while len(dec) < nblk * bs * nb:
            rc, used, got = d.call(b"", 512 * nb, AEC_FLUSH)
            assert rc == AEC_OK
            if not got:
                break
            dec += got
        d.end()
        rc, dec_o, _ = oracle_decode(whole, bps, bs, rsi, flags, nblk * bs * nb)
        assert bytes(dec) == dec_o, (bps, bs, rsi, flags)


def test_streaming_decode_resumes_inside_rsis(api):
    """Streaming decode of streams with full-size RSIs, fed in pieces that end anywhere: every call
    re-indexes from the start of the last incomplete RSI (speculative tables + walker from a bit
    offset inside the staged bytes) and must deliver exactly the samples not delivered before."""
    from libaec_amd.api import Decoder
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
        while pos < len(whole):
            step = int(rng.choice([1, 700, 3000, 20000, 60000]))
            chunk = whole[pos:pos + step]
            off = 0
            while True:
                # never more room than the caller expects in all: a rest-of-segment zero run that
                # closes the short last RSI would otherwise be paid out to the nominal segment end
                # (the reference does the same -- the stream does not say where the data stop)
                room = min(int(rng.choice([nb, 1000 * nb, 1 << 20])), len(want) - len(dec))
                rc, used, got = d.call(chunk[off:], room, AEC_NO_FLUSH)
                assert rc == AEC_OK
                dec += got
                off += used
                if off >= len(chunk) and not got:
                    break
            pos += step
        d.end()
        assert bytes(dec) == want, (bps, bs, rsi, flags)


def test_device_api_offsets_index_and_carry(gpu):
    import torch
    bps, bs, rsi, flags = 16, 16, 128, PP
    data = gen(0, 8 << 20)
    rc, want, _, offs, bits = oracle_encode(data, bps, bs, rsi, flags)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data).cuda()
    d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
    assert tb == bits and d_out[:nbytes].cpu().numpy().tobytes() == want
    got_off = d_off.cpu().numpy().astype(np.uint64)
    assert np.array_equal(got_off[:-1], offs) and got_off[-1] == bits
    nrsi, nblk = codec.rsi_count(data.size), codec.block_count(data.size)
    # RSI-parallel decode from the encoder's offset table
    d_dec, status = codec.decode(d_out, nbytes, d_off, nrsi, nblk)
    assert status == 0 and d_dec.cpu().numpy().tobytes() == data.tobytes()
    # serial index pass over the bare stream finds the same table
    d_idx = torch.zeros(nrsi + 8, dtype=torch.int64, device="cuda")
    d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
    codec.index_async(d_out, nbytes, 0, d_idx, nrsi + 8, d_res)
    res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
    assert res["status"] <= 1 and int(res["n_rsi"]) == nrsi and int(res["end_bit"]) == bits
    assert np.array_equal(d_idx.cpu().numpy()[:nrsi].astype(np.uint64), offs)
    # two batches with (bit offset, k) carried == one stream
    cut = rsi * bs * 2 * 700
    o1, n1, tb1, k1, _ = codec.encode(d_in[:cut].clone())
    o2, n2, tb2, k2, _ = codec.encode(d_in[cut:].clone(), start_bit=tb1 % 8, k_in=k1)
    a = bytearray(o1[: tb1 // 8].cpu().numpy().tobytes())
    b = bytearray(o2[:n2].cpu().numpy().tobytes())
    if tb1 % 8:
        b[0] |= int(o1[tb1 // 8])
    assert bytes(a + b) == want and k2 == k_out


@pytest.mark.parametrize("bps,bs,rsi,flags", [
    (16, 16, 128, PP),            # BASELINE config 2 shape
    (8, 8, 128, PP),              # config 5 shape
    (12, 32, 32, PP | MSB),       # bits not a multiple of 8, short RSIs
    (16, 8, 100, 0),              # no preprocessor (no reference sample), rsi not a multiple of 64
    (4, 16, 64, PP | AEC_RESTRICTED),
    (32, 32, 4096, PP | MSB | SGN),   # BASELINE config 3 shape: RSIs of a megabit -- hop tables
    (16, 64, 256, PP | MSB),          # the reference's own sample shape (src/benc.sh:7)
])
def test_speculative_index_mixed_content(gpu, bps, bs, rsi, flags):
    """Index pass over bare streams whose RSIs are wildly different in size: smooth data (RSIs that
    fit the look-ahead of the speculation windows: table hops), noise (RSIs longer than the
    look-ahead: serial walk), constant stretches (zero-block runs incl. rest-of-segment codes), in
    pieces that do not line up with RSIs.  The offsets must be the encoder's, for several stream
    sizes (one window, several windows, several table chunks) and from a start inside the stream."""
    import torch
    rng = np.random.default_rng(bps * 1000 + bs)
    nb = bytes_per_sample(bps, flags)
    hi = (1 << bps) - 1
    for n_samples in (40_000, 1_500_000, 24_000_000):
        parts, left = [], n_samples
        while left > 0:
            n = int(min(left, rng.integers(200, max(300, n_samples // 6))))
            kind = rng.integers(0, 4)
            if kind == 0:
                v = np.cumsum(rng.integers(-3, 4, n)) + hi // 2          # smooth
            elif kind == 1:
                v = rng.integers(0, hi + 1, n)                           # noise
            elif kind == 2:
                v = np.full(n, int(rng.integers(0, hi + 1)))             # constant
            else:
                v = np.cumsum(rng.integers(-40, 41, n)) + hi // 2        # rougher
            parts.append(np.clip(v, 0, hi))
            left -= n
        vals = np.concatenate(parts)[:n_samples]
        data = np.frombuffer(pack_samples(vals, bps, flags), dtype=np.uint8).copy()
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        d_out, nbytes, tb, _, d_off = codec.encode(d_in)
        nrsi = codec.rsi_count(data.size)
        offs = d_off.cpu().numpy().astype(np.uint64)
        full = (data.size // nb // bs) // rsi                           # RSIs with all their blocks
        for first in (0, full // 3):
            d_idx = torch.zeros(nrsi + 8, dtype=torch.int64, device="cuda")
            d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
            codec.index_async(d_out, nbytes, int(offs[first]), d_idx, nrsi + 8 - first, d_res)
            res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
            assert int(res["status"]) <= 1, (n_samples, first)
            assert full - first <= int(res["n_rsi"]) <= nrsi - first, (n_samples, first, int(res["n_rsi"]))
            got = d_idx.cpu().numpy()[:full - first].astype(np.uint64)
            assert np.array_equal(got, offs[first:full]), (n_samples, first)
            assert int(res["end_bit"]) == tb or int(res["tail_blocks"]) > 0, (n_samples, first)


@pytest.mark.parametrize("kind,bps,bs,rsi,flags,expect_len", [
    (0, 16, 16, 128, PP, 11923045),                       # BASELINE configs 2/4 shape
    (1, 32, 32, 4096, PP | MSB | SGN, 16213210),          # config 3
    (2, 8, 8, 128, PP, 24265016),                         # config 5 kernel shape
])
def test_synthetic_64mib_against_reference_sizes(gpu, kind, bps, bs, rsi, flags, expect_len):
    """64 MiB of each benchmark input: compressed size must equal what the reference produced
     and the bytes must equal the oracle's / reference's."""
    import torch
    data = gen(kind, 64 << 20)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data).cuda()
    d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
    assert nbytes == expect_len
    got = d_out[:nbytes].cpu().numpy().tobytes()
    if have_ref():
        rc, want = ref_encode(data, bps, bs, rsi, flags)
    else:
        rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
    assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
    d_dec, status = codec.decode(d_out, nbytes, d_off, codec.rsi_count(data.size), codec.block_count(data.size))
    assert status == 0 and torch.equal(d_dec, d_in)


def test_full_size_round_trip_config2(gpu):
    """BASELINE config 2 at full size (4 GiB, 16-bit, block 16, rsi 128): too large for the CPU
    oracle in a test, so parity is checked through properties: decode(encode(x)) == x on the
    device, the RSI offset table is strictly increasing and ends at total_bits, and the first
    64 MiB of the stream are byte-identical to the oracle's encoding of the first 64 MiB
    (the stream is causal: a prefix of whole RSIs codes to a prefix of the stream)."""
    import torch
    total = 4 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < 16 << 30:
        pytest.skip("not enough device memory")
    bps, bs, rsi, flags = 16, 16, 128, PP
    data = gen(0, total)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data).cuda()
    d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
    off = d_off.cpu().numpy()
    assert np.all(np.diff(off) > 0) and int(off[-1]) == tb
    pre = 64 << 20
    rc, want, _, o_off, bits = oracle_encode(data[:pre], bps, bs, rsi, flags)
    nfull = bits // 8
    assert d_out[:nfull].cpu().numpy().tobytes() == want[:nfull]
    assert int(off[pre // (rsi * bs * 2)]) == bits
    d_dec, status = codec.decode(d_out, nbytes, d_off, codec.rsi_count(total), codec.block_count(total))
    assert status == 0 and torch.equal(d_dec, d_in)
    del d_dec
    # the RSI starts found again from the stream ALONE (30 spans of window tables, pipelined over two streams):
    # the same table as the encoder's
    nr = codec.rsi_count(total)
    d_idx = torch.zeros(nr + 2, dtype=torch.int64, device=d_in.device)
    d_res = torch.zeros(40, dtype=torch.uint8, device=d_in.device)
    codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
    torch.cuda.synchronize()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
    assert int(res[0]) == nr                                    # n_rsi
    assert torch.equal(d_idx[:nr], d_off[:nr])


def test_segment_parallel_decode(gpu):
    """Decoding one lane per SEGMENT from the encoder's segment table gives the same bytes as
    decoding one lane per RSI -- for large RSIs (config 3 shape), rsi not a multiple of 64, a short
    final RSI, zero runs crossing nothing, signed data."""
    import torch
    rng = np.random.default_rng(31)
    cases = [
        (32, 32, 4096, PP | MSB | SGN, gen(1, 8 << 20)),                     # BASELINE config 3 shape
        (16, 16, 130, PP, gen(0, (16 * 130 * 2) * 37 + 16 * 2 * 5 + 2)),     # 3 segments per RSI, short tail
        (8, 8, 128, PP | MSB, gen(2, 1 << 20)),
        (16, 64, 70, 0, pack_samples(random_walk_samples(rng, 64 * 70 * 5 + 99, 16, 0, scale=0.4, zero_frac=0.6), 16, 0)),
        (12, 16, 300, PP | SGN, pack_samples(random_walk_samples(rng, 16 * 300 * 3, 12, PP | SGN, scale=30.0), 12, PP | SGN)),
    ]
    # short coded data sets on average (the decoder then runs on its half-size ring) with a maximal,
    # uncompressed one every 37th / 5th block: those are decoded a second time after a full refill
    for bps, bs, every in ((16, 16, 37), (32, 32, 5), (8, 8, 11)):
        n = bs * 128 * 40
        v = np.cumsum(rng.integers(-2, 3, n)) + (1 << (bps - 1))
        noise = rng.integers(0, 1 << bps, n, dtype=np.uint64)
        blk = np.arange(n) // bs
        v = np.where(blk % every == every - 1, noise, np.clip(v, 0, (1 << bps) - 1).astype(np.uint64))
        cases.append((bps, bs, 128, PP, pack_samples(v, bps, PP)))
    # high-rate streams (the decoder keeps 4 or 8 loads in flight per block instead of 2): incompressible
    # samples in every block size; and the staging-row shapes of small blocks (8, 16, 32 bytes per block)
    # with block counts that end inside a row
    for bps, bs, rsi in ((16, 16, 128), (16, 32, 64), (16, 64, 70), (32, 64, 33), (8, 32, 65), (24, 16, 50)):
        n = bs * rsi * 9 + bs * 3
        noise = rng.integers(0, 1 << bps, n, dtype=np.uint64)
        cases.append((bps, bs, rsi, PP, pack_samples(noise, bps, PP)))
    for bps, bs, rsi, nblk_total in ((8, 8, 128, 128 * 7 + 61), (8, 8, 3, 3 * 50 + 1), (16, 8, 65, 65 * 9 + 5), (8, 16, 67, 67 * 6 + 3),
                                     (8, 32, 64, 64 * 5 + 1), (16, 16, 128, 128 * 4 + 63)):
        n = bs * nblk_total
        v = np.clip(np.cumsum(rng.integers(-2, 3, n)) + (1 << (bps - 1)), 0, (1 << bps) - 1).astype(np.uint64)
        v[rng.random(n) < 0.3] = 0
        cases.append((bps, bs, rsi, PP, pack_samples(v, bps, PP)))
    for bps, bs, rsi, flags, data in cases:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        nseg = codec.segment_count(data.size)
        d_tab = torch.zeros(nseg * 16, dtype=torch.uint8, device="cuda")
        codec.set_segment_table(d_tab)
        d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
        codec.set_segment_table(None)
        tab = d_tab.cpu().numpy().view(gpu.SEG_ENTRY_DTYPE)
        spr = (rsi + 63) // 64
        assert np.array_equal(tab["bit"][::spr][: len(d_off) - 1], d_off.cpu().numpy()[:-1].astype(np.uint64))
        nrsi, nblk = codec.rsi_count(data.size), codec.block_count(data.size)
        d_ref, st = codec.decode(d_out, nbytes, d_off, nrsi, nblk)
        assert st == 0
        nb = bytes_per_sample(bps, flags)
        d_seg = torch.empty(nblk * bs * nb + 16, dtype=torch.uint8, device="cuda")
        d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
        codec.decode_segments_async(d_out, nbytes, d_tab, nseg, nblk, d_seg, d_res)
        res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
        assert res["status"] == 0, (bps, bs, rsi, flags)
        assert torch.equal(d_seg[: nblk * bs * nb], d_ref), (bps, bs, rsi, flags)
        if not (flags & SGN and bps % 8):         # (signed samples come back sign-extended into their container)
            assert torch.equal(d_ref[: data.size], d_in), (bps, bs, rsi, flags)
        rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert d_out[:nbytes].cpu().numpy().tobytes() == want


def test_device_paths_random_sweep(gpu):
    """Random stream parameters through the DEVICE entry points on inputs large enough for the kernels'
    multi-segment paths (several segments per wave, paired feeds, staging rows, loads in flight sized for the
    coded rate): the stream must be the oracle's, and decoding it per RSI and per segment must both give the
    input back.  AEC_SWEEP_ITERS / AEC_SWEEP_SEED turn it into a soak test."""
    import torch
    rng = np.random.default_rng(int(os.environ.get("AEC_SWEEP_SEED", "4711")))
    for it in range(max(12, int(os.environ.get("AEC_SWEEP_ITERS", "150")) // 6)):
        bps = int(rng.choice([8, 16, 32, 12, 24, 5]))
        bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 3, 64, 65, 128, 130, 300, 4096]))
        flags = PP if rng.random() < 0.8 else 0
        if rng.random() < 0.4:
            flags |= MSB
        if rng.random() < 0.3 and bps in (8, 16, 32):    # (signed samples come back sign-extended into their container)
            flags |= SGN
        nb = bytes_per_sample(bps, flags)
        n = int(rng.integers(1 << 18, 1 << 21)) // (bs * nb) * bs + int(rng.integers(0, bs))
        if it % 6 == 5:
            n *= 20                               # enough segments for a wave to walk several of them
        mode = rng.integers(0, 3)
        if mode == 0:
            vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 2, 40])), zero_frac=float(rng.choice([0.02, 0.5])))
        elif mode == 1:
            lo = -(1 << (bps - 1)) if flags & SGN else 0
            vals = rng.integers(lo, lo + (1 << bps), size=n)
        else:
            vals = np.repeat(rng.integers(0, 1 << min(bps, 6), size=n // 61 + 1), 61)[:n]
        data = np.ascontiguousarray(pack_samples(vals, bps, flags), dtype=np.uint8)
        tag = (it, bps, bs, rsi, flags, n)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        nseg = codec.segment_count(data.size)
        d_tab = torch.zeros(nseg * 16, dtype=torch.uint8, device="cuda")
        codec.set_segment_table(d_tab)
        d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
        codec.set_segment_table(None)
        rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK and d_out[:nbytes].cpu().numpy().tobytes() == want, tag
        nrsi, nblk = codec.rsi_count(data.size), codec.block_count(data.size)
        d_ref, st = codec.decode(d_out, nbytes, d_off, nrsi, nblk)
        assert st == 0, tag
        d_seg = torch.empty(nblk * bs * nb + 16, dtype=torch.uint8, device="cuda")
        d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
        codec.decode_segments_async(d_out, nbytes, d_tab, nseg, nblk, d_seg, d_res)
        assert d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]["status"] == 0, tag
        assert torch.equal(d_seg[: nblk * bs * nb], d_ref), tag
        assert torch.equal(d_ref[: data.size], d_in), tag


def test_pad_rsi_and_restricted_through_abi(api):
    """Decoder-side AEC_PAD_RSI (reference decode.c:407-408) and the restricted code-option sets
    (id_len 1 and 2, reference encode.c:843-851) through the libaec ABI."""
    rng = np.random.default_rng(3)
    bps, bs, rsi, flags = 16, 16, 8, PP
    vals = random_walk_samples(rng, bs * rsi * 5, bps, flags, scale=3.0, zero_frac=0.2)
    data = pack_samples(vals, bps, flags)
    rb = bs * rsi * 2
    stream = b"".join(oracle_encode(data[i:i + rb], bps, bs, rsi, flags)[1] for i in range(0, data.size, rb))
    rc, dec = api.aec_buffer_decode(stream, bps, bs, rsi, flags | 32, data.size)
    assert rc == AEC_OK and dec == data.tobytes()
    for bps in (1, 2, 3, 4):
        fl = PP | AEC_RESTRICTED
        vals = rng.integers(0, 1 << bps, size=5000)
        vals[1000:3000] = vals[1000]
        check_roundtrip(api, f"restricted-{bps}", bps, 16, 10, fl, pack_samples(vals, bps, fl))
    # (the same with streams of megabytes: headers of one or two bits through the window tables of the index pass)
    for bps, rsi in ((2, 64), (4, 128)):
        fl = PP | AEC_RESTRICTED
        vals = np.clip(np.cumsum(rng.integers(-1, 2, size=3_000_000)) + (1 << (bps - 1)), 0, (1 << bps) - 1)
        vals[rng.random(vals.size) < 0.3] = 0
        check_roundtrip(api, f"restricted-{bps}-3M", bps, 16, rsi, fl, pack_samples(vals, bps, fl))


def _check_truncated(dec, dec_o, plain, bs, nb, what):
    """What a stream cut inside a coded data set must give: every sample whose bits arrived, exactly as
    the reference's resumable readers release them (see aec_abi.cpp decode_run / k_decode PARTIAL)."""
    assert dec_o == plain[: len(dec_o)]
    assert dec == dec_o, (what, len(dec), len(dec_o))


def test_corrupt_and_truncated_streams(api):
    """A stream cut inside a coded data set: compared with the oracle's decode of the same truncated
    input (reference decode.c:342-400, 423-460).  A zero-run that overruns its RSI is AEC_DATA_ERROR
    (reference decode.c:543-544)."""
    data = pack_samples(np.arange(4096) * 3 % 4096, 16, PP)
    rc, enc = api.aec_buffer_encode(data, 16, 16, 16, PP)
    rng = np.random.default_rng(11)
    for cut in [len(enc) // 2] + [int(v) for v in rng.integers(1, len(enc), 12)]:
        rc, dec = api.aec_buffer_decode(enc[:cut], 16, 16, 16, PP, data.size)
        rc_o, dec_o, _ = oracle_decode(enc[:cut], 16, 16, 16, PP, data.size)
        assert rc == rc_o == AEC_OK
        _check_truncated(dec, dec_o, data.tobytes(), 16, 2, cut)
    # all-zero blocks in an RSI of 3: ID 0000 + 0 (zero run) + ref 16 bits + fs code "0000001"
    # (fs = 6 -> 6 blocks) overruns the 3-block RSI
    bad = bytes([0b00000000, 0x00, 0x00, 0b00000001, 0x00])
    rc, _ = api.aec_buffer_decode(bad, 16, 16, 3, PP, 4096)
    rc_o, _, _ = oracle_decode(bad, 16, 16, 3, PP, 4096)
    assert rc == rc_o == api.AEC_DATA_ERROR


def test_truncated_streams_release_what_arrived(api):
    """Streams of all code options cut at random bytes, one-shot and fed in two pieces: the samples
    delivered are exactly those the reference's resumable readers release (oracle pinned against the
    reference on this in tests/test_oracle.py::test_truncated_streams_vs_reference)."""
    from libaec_amd.api import Decoder
    rng = np.random.default_rng(2718)
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
            rc_o, dec_o, _ = oracle_decode(enc[:cut], bps, bs, rsi, flags, cap)
            rc, dec = api.aec_buffer_decode(enc[:cut], bps, bs, rsi, flags, cap)
            assert rc == rc_o == AEC_OK and dec == dec_o, (it, bps, bs, rsi, flags, n, cut, len(dec), len(dec_o))
            # the same bytes in two calls: what the first call released early must not come again
            first = int(rng.integers(0, cut + 1))
            d = Decoder(bps, bs, rsi, flags)
            got = bytearray()
            for piece in (enc[:first], enc[first:cut], b""):
                off = 0
                while True:
                    rc, used, out = d.call(piece[off:], cap, AEC_NO_FLUSH if piece else AEC_FLUSH)
                    assert rc == AEC_OK
                    got += out
                    off += used
                    if off >= len(piece) and not out:
                        break
            d.end()
            assert bytes(got) == dec_o, (it, bps, bs, rsi, flags, n, cut, first, len(got), len(dec_o))


@pytest.mark.parametrize("env", [{"AEC_ENC_FUSED": "1"}, {"AEC_ENC_FUSED": "1", "AEC_FUSED_SEGS": "1"},
                                 {"AEC_ENC_FUSED": "1", "AEC_FUSED_SEGS": "2", "AEC_FUSED_PARTS": "3"},
                                 {"AEC_ENC_FUSED": "1", "AEC_FUSED_SEGS": "4", "AEC_FUSED_PARTS": "1"}, {}])
def test_fused_encoder_edges(env):
    """tests/fused_edges.py (streams of zero-block runs whose waves / partitions begin and end inside
    one word, every partition fill, ragged ends) under several geometries of the single-pass encoder
    (opt-in: AEC_ENC_FUSED=1) and once through the default two-pass kernels."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    if env:     # the variant knobs exist in the tuning build only (libaec_amd/csrc/aec_tune.h)
        e["AEC_AMD_LIB"] = os.path.join(ROOT, "libaec_amd", "lib", "tuning", "libaec.so.0")
        assert os.path.exists(e["AEC_AMD_LIB"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fused_edges.py")], env=e, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "fused edges ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_every_index_scheme_on_the_same_streams():
    """tests/cross_scheme.py: ~40 streams across the dispatch thresholds of launch_index (RSIs of 1 .. 4096 blocks, 100 KiB ..
    20 MiB, with and without the preprocessor), each through EVERY scheme that may legally take it -- the tuning build's
    switches take the schemes away one after the other: every bit parsed, regions, phase-locked chains, window tables,
    trunk -- against the oracle encoder's RSI starts and the oracle's bytes: whole, cut short, with the caller's bound in
    front of the end, and resumed inside an RSI."""
    import subprocess
    import sys
    e = dict(os.environ)
    e["AEC_AMD_LIB"] = os.path.join(ROOT, "libaec_amd", "lib", "tuning", "libaec.so.0")
    assert os.path.exists(e["AEC_AMD_LIB"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cross_scheme.py")], env=e, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0 and "cross scheme ok" in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])


@pytest.mark.parametrize("bps,bs,rsi,n_rsi,long_hi", [
    (16, 16, 8, 200, 120),      # coded data sets of up to 1900 bits where the encoder's bound is 277
    (8, 8, 128, 20, 200),
    (32, 32, 16, 40, 400),      # up to 12 800 bits in one coded data set (bound: 1062)
    (16, 64, 4, 60, 150),
])
def test_overlong_coded_data_sets_of_a_foreign_encoder(api, bps, bs, rsi, n_rsi, long_hi):
    """The format does not bound the length of a coded data set -- only sensible encoders do (reference
    decode.c:462-502 decodes fundamental sequences of any length).  The block-parallel decoder keeps a
    bounded look-ahead per lane; a coded data set that does not fit must still come out right."""
    rng = np.random.default_rng(bps + bs)
    id_len = {8: 3, 16: 4, 32: 5}[bps]
    enc = craft_overlong_stream(rng, bps, bs, rsi, n_rsi, id_len, 0.05, long_hi)
    nbytes = n_rsi * rsi * bs * bytes_per_sample(bps, PP)
    rc_o, dec_o, _ = oracle_decode(enc, bps, bs, rsi, PP, nbytes)
    assert rc_o == AEC_OK and len(dec_o) == nbytes
    if have_ref():
        rc_r, dec_r = ref_decode(enc, bps, bs, rsi, PP, nbytes)
        assert rc_r == AEC_OK and dec_r == dec_o
    rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, PP, nbytes)
    assert rc == AEC_OK
    assert dec == dec_o
    # the same through the streaming interface, input and output in pieces
    from libaec_amd.api import Decoder
    d = Decoder(bps, bs, rsi, PP)
    got_all = bytearray()
    pos = 0
    while pos < len(enc):
        chunk = enc[pos:pos + 4093]
        off = 0
        while True:
            rc, used, got = d.call(chunk[off:], 65536, AEC_NO_FLUSH)
            assert rc == AEC_OK
            got_all += got
            off += used
            if off >= len(chunk) and not got:
                break
        pos += 4093
    while len(got_all) < nbytes:
        rc, used, got = d.call(b"", 65536, AEC_FLUSH)
        assert rc == AEC_OK
        if not got:
            break
        got_all += got
    d.end()
    assert bytes(got_all) == dec_o


@pytest.mark.parametrize("bps,bs,rsi,flags,n_rsi,n_chunks", [
    (8, 8, 128, PP, 64, 40),              # the SZIP shape: 64 KiB chunks
    (16, 16, 128, PP, 8, 33),
    (32, 32, 16, PP | MSB | SGN, 4, 17),
    (12, 64, 5, PP, 3, 9),
])
def test_uniform_batch_encode_device_api(gpu, bps, bs, rsi, flags, n_rsi, n_chunks):
    """aec_gpu_encode_uniform_batch_async: n equal chunks as ONE launch set -- every stream must be what the
    oracle makes of its chunk alone (k from 0, zero-padded to a byte), the streams back to back."""
    import torch
    rng = np.random.default_rng(bps * 3 + n_chunks)
    nb = bytes_per_sample(bps, flags)
    chunk_samples = bs * rsi * n_rsi
    vals = random_walk_samples(rng, chunk_samples * n_chunks, bps, flags, scale=2.5, zero_frac=0.2)
    data = pack_samples(vals, bps, flags)
    chunk_bytes = chunk_samples * nb
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
    d_out, rec = codec.encode_uniform_batch(d_in, chunk_bytes, n_chunks)
    out = d_out.cpu().numpy()
    at = 0
    for i in range(n_chunks):
        rc, want, *_ = oracle_encode(data[i * chunk_bytes:(i + 1) * chunk_bytes], bps, bs, rsi, flags)
        assert rc == AEC_OK
        base, bits = int(rec[i, 0]), int(rec[i, 1])
        assert base == at * 8 and (bits + 7) // 8 == len(want), (i, base, bits, len(want))
        assert out[at:at + len(want)].tobytes() == want, i
        at += len(want)


@pytest.mark.parametrize("noise_bits", [16, 13])
def test_index_bridges_rsis_longer_than_the_look_ahead(gpu, noise_bits):
    """A low-entropy stream with long RSIs sprinkled in (each several times the average coded RSI, more than the
    window tables' look-ahead) -- incompressible ones (16 random bits per sample: runs of uncompressed coded data sets
    with the odd split block, found by their headers and bridged from the first span on) and noisy but compressible
    ones (13 random bits: split options of k = 11 / 12, bridged once the walker has met sixteen of them, walked by
    it until then): the RSI starts found from the stream alone must equal the encoder's table either way."""
    import torch
    n = 384 << 20
    free, _ = torch.cuda.mem_get_info()
    if free < 4 << 30:
        pytest.skip("not enough device memory")
    bps, bs, rsi, flags = 16, 16, 128, PP
    data = gen(0, n)
    rsi_bytes = rsi * bs * 2
    nr = n // rsi_bytes
    rng = np.random.default_rng(3)
    for r in rng.choice(nr, size=nr // 100, replace=False):
        noise = rng.integers(0, 1 << noise_bits, rsi_bytes // 2, dtype=np.uint16)
        data[r * rsi_bytes:(r + 1) * rsi_bytes] = noise.view(np.uint8)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data).cuda()
    d_out, nbytes, tb, _, d_off = codec.encode(d_in)
    d_idx = torch.zeros(nr + 2, dtype=torch.int64, device=d_in.device)
    d_res = torch.zeros(40, dtype=torch.uint8, device=d_in.device)
    codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
    torch.cuda.synchronize()
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
    assert int(res[0]) == nr
    assert torch.equal(d_idx[:nr], d_off[:nr])
    d_dec, status = codec.decode(d_out, nbytes, d_idx, nr, codec.block_count(n))
    assert status == 0 and torch.equal(d_dec, d_in)


def test_random_sweep_of_the_bare_stream_path(gpu):
    """tests/fuzz_index_gpu.py, 30 cases: random parameters (bits per sample 8..32 in their containers, block sizes,
    RSIs of 1..4096 blocks, byte order, signedness, with and without the preprocessor), random-walk data of several
    entropies with zero stretches, jumps and incompressible pieces, 1..48 MiB: the stream against the oracle's, the
    RSI starts found from the stream alone against the encoder's table, the bytes decoded from the found starts
    against the oracle's."""
    import argparse
    import fuzz_index_gpu
    assert fuzz_index_gpu.run(argparse.Namespace(cases=30, seed=5, only="", time=False)) == 0


def test_random_sweep_of_corrupted_streams(api):
    """tests/fuzz_corrupt_gpu.py, 60 cases: bit flips, cuts with garbage tails and overwritten stretches in valid
    streams through aec_buffer_decode.  Where the oracle (the reference's behaviour: it decodes until the output
    is full and only then looks no further) returns AEC_OK the product returns the same bytes -- also when the
    damage made coded data sets the walker or the decoder must refuse BEHIND the output asked for, and when the
    predictor state has left the sample range by the time the input ends inside a coded data set -- and where it
    returns AEC_DATA_ERROR, so does the product."""
    import argparse
    import fuzz_corrupt_gpu
    assert fuzz_corrupt_gpu.run(argparse.Namespace(cases=60, seed=13)) == 0


def test_random_sweep_of_the_streaming_calls(api):
    """tests/fuzz_stream_gpu.py, 40 cases: aec_encode / aec_decode in random pieces (7 bytes .. 1 MiB of input and of
    room per call), the same calls against the compiled reference: same coded stream; same decoded bytes and return
    code for valid and damaged streams.  Where the reference's own answer depends on the room it is handed (a zero
    run past the end of its RSI is only refused when the room holds the whole run, decode.c:543-544) the product
    gives the answer of the reference with ample room, in pieces as in one call."""
    import argparse
    import fuzz_stream_gpu
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    assert fuzz_stream_gpu.run(argparse.Namespace(cases=40, seed=1, dump="", ref_only=False, big=False)) == 0


def test_random_sweep_of_the_streaming_calls_big(api):
    """The same sweep with 1 .. 4 M samples per case in pieces of 4 KiB .. 3 MiB (several decode batches per stream, walks
    that resume inside an RSI): seed 504 is the one that found round 4's wrong output behind a bounded index pass over
    short RSIs (it ran by hand only)."""
    import argparse
    import fuzz_stream_gpu
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    assert fuzz_stream_gpu.run(argparse.Namespace(cases=12, seed=504, dump="", ref_only=False, big=True)) == 0
    assert fuzz_stream_gpu.run(argparse.Namespace(cases=8, seed=9001, dump="", ref_only=False, big=True)) == 0


def test_large_decodes_of_streams_with_short_rsis(api):
    """48 MiB and more with RSIs of 1 .. 32 blocks through aec_buffer_decode and through the streaming calls: several
    batches of the phase-locked index pass, each but the first bounded in the middle of the stream or resumed where
    the one in front ended (one-shot: output-bounded batches; streaming: input in pieces of 3 MiB)."""
    import fuzz_stream_gpu
    for bps, bs, rsi, kind, n in ((8, 8, 4, 2, 48 << 20), (16, 16, 16, 0, 64 << 20), (8, 8, 1, 2, 48 << 20)):
        data = gen(kind, n)
        rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, PP)
        assert rc == AEC_OK
        rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, PP, n)
        assert rc == AEC_OK and dec == data.tobytes(), (bps, bs, rsi, "one shot")
        rc, got = fuzz_stream_gpu.drive(api.library(), "decode", enc, (bps, bs, rsi, PP), [(3 << 20, 1 << 30)], n)
        assert rc == AEC_OK and got == data.tobytes(), (bps, bs, rsi, "streamed")


def test_streams_the_first_scheme_gives_up(gpu):
    """The every-bit scheme as the fallback (aec_idx.hip: small_fallback_plan): streams of tests/fuzz_index_gpu.py (seed 5001)
    whose entries by plausibility are judged wrong (cases 22, 212: data in blocks of 64 with RSIs of 5 and 1 blocks -- 61
    and 122 ms over the 64 agreeing chains until round 5) and whose window tables resolve next to nothing (47, 94: the
    walker gives up after 2048 blocks walked serially -- 15 and 54 ms until round 5): RSI starts against the encoder's
    table, stream and decoded bytes against the oracle."""
    import argparse
    import fuzz_index_gpu
    assert fuzz_index_gpu.run(argparse.Namespace(cases=213, seed=5001, only="22,47,94,212", time=False)) == 0


def test_streams_with_short_rsis(api, gpu):
    """RSIs of 1 .. 32 blocks (narrow SZIP scan lines): no table scheme of the index pass applies -- its chains parse
    without reference samples and every few coded data sets hold one -- and until round 4 every RSI fell to the serial
    walker (16 MiB of 8-bit data with rsi 1: 3.2 s; the reference on one core: 0.05 s).  Now phase-locked chains find
    the RSI starts (aec_idx.hip: launch_index_locked).  Offsets against the encoder's table, decoded bytes against the
    oracle for whole, cut and garbage-tailed streams, and a loose bound on the time."""
    import time
    import torch
    rng = np.random.default_rng(32)
    for bps, bs, rsi, kind in ((8, 8, 1, 2), (8, 8, 5, 2), (16, 16, 3, 0), (16, 16, 16, 0), (16, 32, 32, 0), (8, 16, 8, 2)):
        n = 6 << 20
        data = gen(kind, n)
        flags = PP
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        d_out, nbytes, tb, _, d_off = codec.encode(d_in)
        nr = codec.rsi_count(n)
        d_idx = torch.zeros(nr + 2, dtype=torch.int64, device="cuda")
        d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
        codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
        whole = codec.block_count(n) // rsi              # (a short last RSI counts as the trailing incomplete one)
        assert int(res[0]) in (nr, whole) and torch.equal(d_idx[:whole], d_off[:whole]), (bps, bs, rsi)
        # the PATH, not a wall-clock bound (times: tests/bench_short_rsi.py): phase-locked chains, also
        # for a walk that resumes inside an RSI (streaming callers whose input arrives in pieces)
        hint = nbytes * 8 // max(nr, 1)
        # (streams of at most 2 MiB walked from an RSI start: every bit parsed, test_small_streams_every_bit_parsed)
        assert gpu.index_scheme(bps, bs, rsi, flags, nbytes, hint, 0) == (4 if nbytes <= 2 << 20 else 1), (bps, bs, rsi)
        # (round 6: a walk that resumes inside an RSI is the every-bit scheme's as well, up to 2 MiB)
        assert gpu.index_scheme(bps, bs, rsi, flags, nbytes // 2, hint, 1) == (4 if nbytes // 2 <= 2 << 20 else 1), (bps, bs, rsi)
        # a caller's bound in the middle of the stream (what every batch but the last of a large decode asks for): ends
        # on the start of RSI number `bound`, which ONE region delivers (every region behind it is past the bound too)
        bound = whole // 3 + 1
        d_idx.zero_()
        codec.index_async(d_out, nbytes, 0, d_idx, bound, d_res)
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
        assert int(res[0]) == bound and int(res[1]) == 0 and int(res[2]) == int(d_off[bound]), (bps, bs, rsi, res[:3])
        assert torch.equal(d_idx[:bound], d_off[:bound]), (bps, bs, rsi)
        enc = d_out[:nbytes].cpu().numpy().tobytes()
        # ... and the streaming calls with little room per call: several batches, each resuming where the last ended
        import fuzz_stream_gpu
        rc, got = fuzz_stream_gpu.drive(api.library(), "decode", enc, (bps, bs, rsi, flags), [(1 << 30, 1 << 20)], n)
        assert rc == AEC_OK and got == data.tobytes(), (bps, bs, rsi, len(got))
        # ... and with the INPUT in pieces (1 MiB, then odd sizes): a batch then ends wherever the input does, the next
        # walk resumes INSIDE an RSI (start_block != 0), which took the serial walker until round 5
        rc, got = fuzz_stream_gpu.drive(api.library(), "decode", enc, (bps, bs, rsi, flags),
                                        [(1 << 20, 1 << 30), (300007, 1 << 30), (1 << 20, 1 << 30)], n)
        assert rc == AEC_OK and got == data.tobytes(), (bps, bs, rsi, len(got))
        for name, stream in (("whole", enc), ("cut", enc[: int(len(enc) * 0.61)]),
                             ("cut + garbage", enc[: len(enc) // 3] + bytes(rng.integers(0, 256, 300, dtype=np.uint8).tolist())),
                             ("garbage tail", enc + bytes(rng.integers(0, 256, 100, dtype=np.uint8).tolist()))):
            rc_o, dec_o, _ = oracle_decode(stream, bps, bs, rsi, flags, n)
            rc_p, dec_p = api.aec_buffer_decode(stream, bps, bs, rsi, flags, n)
            assert rc_p == rc_o, (bps, bs, rsi, name, rc_p, rc_o)
            if rc_o == AEC_OK:
                assert dec_p == dec_o, (bps, bs, rsi, name)


def test_long_coded_data_sets_in_short_rsis(api, gpu, typical_rz):
    """The shape of the reference's sample file (16-bit, blocks of 64, rsi 256: coded data sets of ~650 bits, RSIs of
    ~180 kbit): a chain needs longer than an RSI to find the true one, so neither table scheme of the index pass
    applies and until round 5 the trunk walked such streams at 1 GiB per 110 ms.  Now the entries of the regions are
    guessed by the PLAUSIBILITY of the options along a chain (aec_idx.hip: k_lock_guess_p) and checked, repaired and
    judged by the kernels of the phase-locked scheme.  Offsets against the encoder's table; decoded bytes against the
    oracle for whole, cut and garbage-tailed streams and for input in pieces (walks that resume inside an RSI); and a
    stream whose options say nothing (a third of its blocks zero, the others noise of one size), where the guesses are
    judged wrong and the trunk takes over: same results."""
    import torch
    import fuzz_stream_gpu
    rng = np.random.default_rng(256)
    flags = PP | MSB
    rc, one = api.aec_buffer_decode(typical_rz, 16, 64, 256, flags, 1 << 20)
    assert rc == AEC_OK
    tiles = np.tile(np.frombuffer(one, dtype=np.uint8), 8)
    fl2 = PP | MSB | SGN
    walk = np.frombuffer(pack_samples(random_walk_samples(rng, 4 << 20, 16, fl2, scale=8.0, zero_frac=0.3), 16, fl2), dtype=np.uint8)
    for name, bps, bs, rsi, fl, data in (("sample file x 8", 16, 64, 256, flags, tiles),
                                         ("zero blocks and noise", 16, 32, 256, fl2, walk)):
        n = data.size
        codec = gpu.Codec(bps, bs, rsi, fl)
        d_in = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        d_out, nbytes, tb, _, d_off = codec.encode(d_in)
        nr = codec.rsi_count(n)
        hint = nbytes * 8 // max(nr, 1)
        # (round 6: up to 4 MiB of such a stream the every-bit scheme, piece by piece -- the guesses alone take as long)
        assert gpu.index_scheme(bps, bs, rsi, fl, nbytes, hint, 0) == (4 if nbytes * 8 <= 1 << 25 else 1), name
        d_idx = torch.zeros(nr + 2, dtype=torch.int64, device="cuda")
        d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
        codec.index_async(d_out, nbytes, 0, d_idx, nr, d_res)
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
        whole = codec.block_count(n) // rsi
        assert int(res[0]) in (nr, whole) and torch.equal(d_idx[:whole], d_off[:whole]), name
        bound = whole // 3 + 1
        d_idx.zero_()
        codec.index_async(d_out, nbytes, 0, d_idx, bound, d_res)
        torch.cuda.synchronize()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
        assert int(res[0]) == bound and int(res[1]) == 0 and int(res[2]) == int(d_off[bound]), (name, res[:3])
        assert torch.equal(d_idx[:bound], d_off[:bound]), name
        enc = d_out[:nbytes].cpu().numpy().tobytes()
        rc_o, enc_o, *_ = oracle_encode(data, bps, bs, rsi, fl)
        assert rc_o == AEC_OK and enc == enc_o, name
        rc, got = fuzz_stream_gpu.drive(api.library(), "decode", enc, (bps, bs, rsi, fl),
                                        [(1 << 20, 1 << 30), (300007, 1 << 30), (1 << 20, 1 << 30)], n)
        assert rc == AEC_OK and got == data.tobytes(), (name, len(got))
        for what, stream in (("whole", enc), ("cut", enc[: int(len(enc) * 0.61)]),
                             ("cut + garbage", enc[: len(enc) // 3] + bytes(rng.integers(0, 256, 300, dtype=np.uint8).tolist())),
                             ("garbage tail", enc + bytes(rng.integers(0, 256, 100, dtype=np.uint8).tolist()))):
            rc_o, dec_o, _ = oracle_decode(stream, bps, bs, rsi, fl, n)
            rc_p, dec_p = api.aec_buffer_decode(stream, bps, bs, rsi, fl, n)
            assert rc_p == rc_o, (name, what, rc_p, rc_o)
            if rc_o == AEC_OK:
                assert dec_p == dec_o, (name, what)


def test_small_streams_every_bit_parsed(api, gpu):
    """A chunk of a dataset per call (reference src/sz_compat.c:239): streams of at most 2 MiB with RSIs of at most 64
    blocks are parsed at EVERY bit and the chain of RSI starts is ranked by pointer doubling (aec_idx.hip:
    launch_index_small) -- no preprocessor, zero-block runs to the end of a segment, a short last RSI, 8- to 32-bit
    samples.  Offsets against the encoder's table, a caller's bound in the middle of the stream, decoded bytes against
    the oracle for whole, cut and garbage-tailed streams."""
    import torch
    rng = np.random.default_rng(64)
    shapes = [(8, 8, 1, PP, 65536), (8, 8, 4, PP, 65536), (8, 16, 1, 0, 200000), (16, 16, 16, PP | MSB, 300001),
              (16, 64, 1, 0, 100000), (32, 16, 5, PP, 90000), (32, 32, 64, PP | SGN, 250000), (24, 8, 33, PP | MSB, 70001),
              (12, 32, 3, PP, 1 << 20), (16, 8, 64, PP, 3), (8, 8, 64, 0, 1 << 19),
              # the hops: RSIs of more than 16 blocks; hops of hops: more than 256 (where the window tables do not serve)
              (8, 8, 128, PP, 65536), (16, 16, 256, 0, 40000), (32, 32, 4096, PP | MSB, 300000), (16, 8, 1024, PP, 200000),
              # without the preprocessor piece by piece: a stream of 6 MB is three pieces of 16 Mbit
              (16, 16, 16, 0, 3 << 20), (16, 16, 200, 0, (3 << 20) + 77)]
    for bps, bs, rsi, flags, n in shapes:
        vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 3.0, 60.0])),
                                   zero_frac=float(rng.choice([0.0, 0.3, 0.7])))
        raw = np.frombuffer(pack_samples(vals, bps, flags), dtype=np.uint8)
        nb = bytes_per_sample(bps, flags)
        rc, enc, _, offs, _ = oracle_encode(raw, bps, bs, rsi, flags)
        assert rc == AEC_OK
        nblk = (n + bs - 1) // bs
        out = nblk * bs * nb
        assert gpu.index_scheme(bps, bs, rsi, flags, len(enc), 0, 0) == (4 if len(enc) >= 8 else 0), (bps, bs, rsi)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(np.frombuffer(enc, dtype=np.uint8).copy()).cuda()
        d_off = torch.from_numpy(offs.astype(np.int64)).cuda()          # the oracle's table of RSI starts
        nr = len(offs)
        whole = nblk // rsi
        d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
        for bound in (nr, max(1, whole // 3 + 1), 1):
            d_idx = torch.zeros(nr + 2, dtype=torch.int64, device="cuda")
            codec.index_async(d_in, len(enc), 0, d_idx, bound, d_res)
            torch.cuda.synchronize()
            res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.uint64)
            got = min(bound, whole)
            if bound < whole:
                assert int(res[0]) == bound and int(res[1]) == 0 and int(res[2]) == int(d_off[bound]), (bps, bs, rsi, bound, res[:3])
            else:
                assert int(res[0]) in (nr, whole), (bps, bs, rsi, bound, res[:3])
            assert torch.equal(d_idx[:got], d_off[:got]), (bps, bs, rsi, bound)
        for what, stream in (("whole", enc), ("cut", enc[: int(len(enc) * 0.61)]),
                             ("cut + garbage", enc[: len(enc) // 3] + bytes(rng.integers(0, 256, 300, dtype=np.uint8).tolist())),
                             ("garbage tail", enc + bytes(rng.integers(0, 256, 100, dtype=np.uint8).tolist()))):
            # (garbage behind a short last RSI, room for exactly the samples: the reference stops with the room and
            # returns AEC_OK; the oracle restates the decoder with ample room and meets the garbage -- the compiled
            # reference is the arbiter where the two differ, as in tests/fuzz_corrupt_gpu.py)
            if have_ref():
                rc_o, dec_o = ref_decode(stream, bps, bs, rsi, flags, out)
            elif what == "garbage tail" and nblk % rsi:
                continue
            else:
                rc_o, dec_o, _ = oracle_decode(stream, bps, bs, rsi, flags, out)
            rc_p, dec_p = api.aec_buffer_decode(stream, bps, bs, rsi, flags, out)
            assert rc_p == rc_o, (bps, bs, rsi, what, rc_p, rc_o)
            if rc_o == AEC_OK:
                assert dec_p == dec_o, (bps, bs, rsi, what)


def test_large_one_shot_decode_of_damaged_streams(api):
    """aec_buffer_decode of 160 MiB runs as pipelined batches (index pass on a piece of the stream, copy-out of one batch
    beside the kernels of the next).  Damage in the first, a middle and the last batch, a cut
    stream and a garbage tail must come out as from the oracle: same return code, and the same bytes where it is
    AEC_OK (flipped bits can leave a stream that still decodes)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 4 << 30:
        pytest.skip("not enough device memory")
    bps, bs, rsi, flags = 16, 16, 128, PP
    n = 160 << 20
    data = gen(0, n)
    rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, flags)
    assert rc == AEC_OK
    rng = np.random.default_rng(160)
    cases = [("intact", bytes(enc))]
    for name, where in (("first batch", 0.1), ("middle batch", 0.45), ("last batch", 0.93)):
        b = bytearray(enc)
        at = int(len(b) * where)
        b[at:at + 40] = bytes(rng.integers(0, 256, 40, dtype=np.uint8).tolist())
        cases.append((f"40 bytes overwritten in the {name}", bytes(b)))
    cases.append(("cut + garbage", bytes(enc[: int(len(enc) * 0.6)]) + bytes(rng.integers(0, 256, 50, dtype=np.uint8).tolist())))
    cases.append(("garbage tail", bytes(enc) + bytes(rng.integers(0, 256, 200, dtype=np.uint8).tolist())))
    for name, stream in cases:
        rc_o, dec_o = ref_decode(stream, bps, bs, rsi, flags, n) if have_ref() else oracle_decode(stream, bps, bs, rsi, flags, n)[:2]
        rc_p, dec_p = api.aec_buffer_decode(stream, bps, bs, rsi, flags, n)
        assert rc_p == rc_o, (name, rc_p, rc_o)
        if rc_o == AEC_OK:
            assert dec_p == dec_o, name
        if name in ("intact", "garbage tail"):
            assert rc_p == AEC_OK and np.array_equal(np.frombuffer(dec_p, dtype=np.uint8), data), name


def test_flush_calls_after_the_stream_is_complete(api):
    """A caller whose buffer came back FULL from the call that completed the stream calls aec_encode(AEC_FLUSH) once
    more.  The reference then writes the byte it still holds again (encode.c:686-695 has no guard for a flush that is
    complete): its stream ends with the last byte twice -- and WHICH calls do that depends on how its output happens
    to fall onto the caller's buffers.  The product hands output out in batches (include/libaec.h), so it cannot repeat
    that byte in the same calls and does not repeat it at all: a complete stream stays complete (deliberate deviation,
    DESIGN.md section 1; found by tests/fuzz_stream_gpu.py --seed 43, case 113)."""
    import fuzz_stream_gpu
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(113)
    for bps, bs, rsi, flags in ((8, 64, 128, 0), (16, 16, 16, PP)):
        vals = random_walk_samples(rng, 12000, bps, flags, scale=3.0, zero_frac=0.2, jump_frac=0.002)
        data = pack_samples(vals, bps, flags)
        params = (bps, bs, rsi, flags)
        rc, whole = fuzz_stream_gpu.drive(ref_lib(), "encode", data, params, [(1 << 30, 1 << 20)], 1 << 20)
        assert rc == 0
        plan = [(1 << 30, len(whole))]                     # the whole stream in one buffer that comes back full
        rc_r, enc_r = fuzz_stream_gpu.drive(ref_lib(), "encode", data, params, plan, 1 << 20)
        rc_p, enc_p = fuzz_stream_gpu.drive(api.library(), "encode", data, params, plan, 1 << 20)
        assert enc_r == whole + whole[-1:], "the reference no longer repeats its last byte"
        assert (rc_p, enc_p) == (0, whole)


def test_wave_per_rsi_decoder_edges(api, gpu):
    """k_decode_wave (a wavefront per RSI, a lane per block; chosen for at most 4096 RSIs of at least 16 blocks) against the
    ORACLE's bytes on what its three phases can get wrong: the walk -- zero runs longer than a round of 64 blocks, rest-of-
    segment codes, blocks of 64 32-bit samples (coded data sets beyond the 2048-bit register window: the sequential reader),
    RSIs longer than the wavefront's window of the stream (4096 blocks); the predictor -- data that clips at both ends of the
    range in most blocks (the lane-against-lane check has to iterate), signed samples, no preprocessor; the store -- 3-byte
    containers, a short last RSI; and the device API with the encoder's offset table as well as the ABI."""
    import torch
    rng = np.random.default_rng(55)

    def clipping(n, bps, signed):
        lo, hi = (-(1 << (bps - 1)), (1 << (bps - 1)) - 1) if signed else (0, (1 << bps) - 1)
        v = np.cumsum(rng.integers(-3, 4, n) * max(1, (hi - lo) // 40))
        return np.clip(v - v.min() // 2 + lo, lo, hi)

    def zero_runs(n, bps):
        v = np.cumsum(rng.integers(-2, 3, n)) + (1 << (bps - 1))
        for _ in range(12):
            a = int(rng.integers(0, n - 5000))
            v[a:a + int(rng.integers(200, 5000))] = v[a]                    # constant stretches: zero runs, ROS codes
        return np.clip(v, 0, (1 << bps) - 1)

    cases = [
        ("clip u16", 16, 16, 128, PP, pack_samples(clipping(16 * 128 * 9 + 37, 16, False), 16, PP)),
        ("clip s8", 8, 8, 64, PP | SGN, pack_samples(clipping(8 * 64 * 40 + 3, 8, True), 8, PP | SGN)),
        ("clip s32 msb", 32, 32, 96, PP | SGN | MSB, pack_samples(clipping(32 * 96 * 7, 32, True), 32, PP | SGN | MSB)),
        ("zero runs", 16, 16, 300, PP, pack_samples(zero_runs(16 * 300 * 11 + 100, 16), 16, PP)),
        ("zero runs 8", 8, 8, 1000, PP, pack_samples(zero_runs(8 * 1000 * 6, 8), 8, PP)),
        ("all zero", 16, 16, 128, PP, np.zeros(16 * 128 * 2 * 5, dtype=np.uint8)),
        ("block 64 x 32 bit", 32, 64, 64, PP, gen(1, 64 * 64 * 4 * 6 + 64 * 4 * 3)),
        ("rsi 4096", 16, 16, 4096, PP, gen(0, 16 * 4096 * 2 * 3 + 16 * 2 * 100)),
        ("rsi 4096 noise", 16, 32, 4096, PP, rng.integers(0, 256, 32 * 4096 * 2 * 2, dtype=np.uint8)),
        ("24 bit 3 byte", 24, 16, 128, PP | AEC_DATA_3BYTE, rng.integers(0, 40, 16 * 128 * 3 * 7, dtype=np.uint8)),
        ("no preprocessor", 16, 16, 128, 0, gen(0, 16 * 128 * 2 * 9)),
        ("ramp", 16, 16, 128, PP, pack_samples(np.arange(16 * 128 * 30) % 65536, 16, PP)),
    ]
    for name, bps, bs, rsi, flags, data in cases:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        nb = bytes_per_sample(bps, flags)
        data = data[: data.size - data.size % nb]
        rc, enc, _, offs, bits = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK, name
        nblk = (data.size // nb + bs - 1) // bs
        rc_o, dec_o, _ = oracle_decode(enc, bps, bs, rsi, flags, nblk * bs * nb)
        assert rc_o == AEC_OK, name
        # the ABI (index pass + decode)
        rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, flags, nblk * bs * nb)
        assert rc == AEC_OK and dec == dec_o, (name, "ABI")
        # the device API with the oracle's offsets (the decoder alone)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_enc = torch.from_numpy(np.frombuffer(enc + b"\0" * 64, dtype=np.uint8).copy()).cuda()
        nrsi = (nblk + rsi - 1) // rsi
        d_off = torch.from_numpy(np.concatenate([np.asarray(offs, dtype=np.uint64), np.array([bits], dtype=np.uint64)])
                                 .astype(np.int64)).cuda()
        d_dec, st = codec.decode(d_enc, len(enc), d_off, nrsi, nblk)
        assert st == 0 and d_dec[: nblk * bs * nb].cpu().numpy().tobytes() == dec_o, (name, "device API")
        # ... and cut inside a coded data set: what the reference releases of it (decode.c:423-460)
        cut = enc[: max(1, int(len(enc) * 0.77))]
        rc_o, dec_c, _ = oracle_decode(cut, bps, bs, rsi, flags, nblk * bs * nb)
        rc, dec = api.aec_buffer_decode(cut, bps, bs, rsi, flags, nblk * bs * nb)
        assert rc == rc_o and dec == dec_c, (name, "cut", rc, rc_o, len(dec), len(dec_c))


def test_degenerate_inputs(api):
    """Streams whose coded data sets are as short or as regular as the format allows (tests/bench_degenerate.py): all
    zeros and constants (an RSI of 128 blocks is two coded data sets: the phase-locked index), ramps (every coded data set
    the same: a periodic stream on which chains from different places never meet; the dense fallback tables), a sawtooth,
    noise in the low bits -- 8 MiB each through aec_buffer_encode / aec_buffer_decode against the oracle."""
    n = 8 << 20
    rng = np.random.default_rng(9)
    for bps, bs, rsi in ((16, 16, 128), (8, 8, 128), (16, 32, 64)):
        dt = np.dtype("<u2") if bps == 16 else np.dtype(np.uint8)
        m = n // dt.itemsize
        shapes = {
            "zeros": np.zeros(m, dtype=dt),
            "constant": np.full(m, 1000 if bps == 16 else 100, dtype=dt),
            "ramp": (np.arange(m) % (1 << bps)).astype(dt),
            "ramp by 3": ((3 * np.arange(m)) % (1 << bps)).astype(dt),
            "sawtooth": (np.arange(m) % 37).astype(dt),
            "noise in 2 low bits": (rng.integers(0, 4, m, dtype=np.uint64) + (1 << (bps - 1))).astype(dt),
        }
        for name, arr in shapes.items():
            data = arr.view(np.uint8)
            rc, enc = api.aec_buffer_encode(data, bps, bs, rsi, PP)
            rc_o, enc_o, *_ = oracle_encode(data, bps, bs, rsi, PP)
            assert rc == AEC_OK and enc == enc_o, (name, bps, bs, rsi)
            rc, dec = api.aec_buffer_decode(enc, bps, bs, rsi, PP, n)
            assert rc == AEC_OK and dec == data.tobytes(), (name, bps, bs, rsi)


def test_bare_stream_decodes_by_segments(gpu):
    """A bare stream with long RSIs (BASELINE config 3 shape: 64 segments per RSI): the index pass leaves the segment
    starts beside the RSI starts (aec_gpu_index_segments_async; the reference has no entry points inside an RSI,
    src/decode.c:402-421) and the decoder takes a lane per segment, the sample in front of each from a summing pass
    (inside the range the inverse predictor of decode.c:96-134 is a running sum).  Output = what a lane per RSI gives =
    the input; also for data that CLIPS at the ends of the range (those RSIs go back to a lane per RSI), a short last
    RSI, rsi not a multiple of 64, no preprocessor, and the reference's sample shape (4 segments per RSI)."""
    import torch
    rng = np.random.default_rng(77)

    def clipping(n, bps, signed):
        # a walk that keeps running into both ends of the range: the predictor's one-sided branch (decode.c:96-134)
        lo, hi = (-(1 << (bps - 1)), (1 << (bps - 1)) - 1) if signed else (0, (1 << bps) - 1)
        v = np.cumsum(rng.integers(-40, 41, n) * (1 << max(0, bps - 12)))
        v = np.clip(v - v.min() // 2 + lo, lo, hi)
        v[: n // 3] = np.clip(v[: n // 3], lo + (hi - lo) // 4, hi)          # (a stretch that stays inside)
        return v

    cases = [
        (32, 32, 4096, PP | MSB | SGN, gen(1, 24 << 20)),                                     # config 3 shape
        (32, 32, 4096, PP | MSB | SGN, gen(1, (3 << 20) + 32 * 4 * 70 + 4)),                 # ... with a short last RSI
        (32, 32, 4096, PP | SGN, pack_samples(clipping(32 * 4096 * 6, 32, True), 32, PP | SGN)),
        (16, 16, 1000, PP, pack_samples(clipping(16 * 1000 * 9 + 5, 16, False), 16, PP)),
        (16, 16, 1000, PP, gen(0, 16 * 1000 * 2 * 40)),                                       # zero-block runs, rest-of-segment codes
        (16, 64, 256, PP | MSB, gen(0, 8 << 20)),                                             # 4 segments per RSI
        (8, 8, 300, PP, gen(2, 8 * 300 * 50 + 8 * 17)),
        (16, 32, 512, 0, gen(0, 4 << 20)),                                                    # no preprocessor
    ]
    for bps, bs, rsi, flags, data in cases:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        codec = gpu.Codec(bps, bs, rsi, flags)
        d_in = torch.from_numpy(data).cuda()
        d_out, nbytes, tb, k_out, d_off = codec.encode(d_in)
        nrsi, nblk = codec.rsi_count(data.size), codec.block_count(data.size)
        nb = bytes_per_sample(bps, flags)
        d_ref, st = codec.decode(d_out, nbytes, d_off, nrsi, nblk)                            # a lane per RSI
        assert st == 0
        spr = codec.segments_per_rsi()
        assert spr == (rsi + 63) // 64
        for with_record in (False, True):
            d_idx = torch.zeros(nrsi + 2, dtype=torch.int64, device="cuda")
            d_sb = torch.zeros((nrsi + 2) * spr, dtype=torch.int64, device="cuda")
            d_ires = torch.zeros(40, dtype=torch.uint8, device="cuda")
            d_res = torch.zeros(40, dtype=torch.uint8, device="cuda")
            d_dec = torch.zeros(nrsi * rsi * bs * nb + 64 * bs * nb, dtype=torch.uint8, device="cuda")
            codec.index_segments_async(d_out, nbytes, 0, d_idx, d_sb, nrsi + 1, d_ires)
            codec.decode_bare_async(d_out, nbytes, d_idx, d_sb, nrsi + 1 if with_record else nrsi, nblk,
                                    d_ires if with_record else None, d_dec, d_res)
            ires = d_ires.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
            res = d_res.cpu().numpy().view(gpu.DEC_RESULT_DTYPE)[0]
            assert int(ires["n_rsi"]) * rsi + int(ires["tail_blocks"]) == nblk, (bps, bs, rsi, flags, ires)
            assert torch.equal(d_idx[:nrsi], d_off[:nrsi])
            assert res["status"] == 0, (bps, bs, rsi, flags, res)
            assert torch.equal(d_dec[: nblk * bs * nb], d_ref[: nblk * bs * nb]), (bps, bs, rsi, flags, with_record)
            # ... and against the ORACLE's bytes (the CPU restatement of the reference decoding the same stream): the
            # comparison above is the product against itself
            if data.size <= (8 << 20) or with_record:
                stream = d_out[:nbytes].cpu().numpy().tobytes()
                rc_o, dec_o, _ = oracle_decode(stream, bps, bs, rsi, flags, nblk * bs * nb)
                assert rc_o == AEC_OK and d_dec[: nblk * bs * nb].cpu().numpy().tobytes() == dec_o, (bps, bs, rsi, flags)
        if not (flags & SGN and bps % 8):
            assert torch.equal(d_ref[: data.size], d_in), (bps, bs, rsi, flags)
