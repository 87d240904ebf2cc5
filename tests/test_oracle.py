"""Pins the oracle (oracle/aec_oracle.c): against the reference's shipped known-answer file,
the golden vectors the reference produced (tests/golden), the option-ID assertions of
reference tests/check_code_options.c:25-31, and -- when oracle/_ref is present -- a
randomized differential sweep against the reference itself."""
import hashlib

import numpy as np
import pytest

from helpers import (AEC_CONF_ERROR, AEC_DATA_3BYTE, AEC_DATA_MSB, AEC_DATA_PREPROCESS,
                     AEC_DATA_SIGNED, AEC_MEM_ERROR, AEC_NOT_ENFORCE, AEC_OK, AEC_RESTRICTED,
                     AEC_STREAM_ERROR, OPT_SE, OPT_SPLIT, OPT_UNCOMP, OPT_ZERO, bytes_per_sample,
                     have_ref, id_len_of, oracle_decode, oracle_encode, pack_samples,
                     random_walk_samples, ref_decode, ref_encode, unpack_samples)

PP, MSB = AEC_DATA_PREPROCESS, AEC_DATA_MSB


def test_typical_rz_known_answer(typical_rz):
    """reference data/typical.rz = `aec -n16 -j64 -r256 -m` + preprocess (src/benc.sh:7)."""
    assert hashlib.sha256(typical_rz).hexdigest().startswith("16a7f994a672")
    rc, dec, used = oracle_decode(typical_rz, 16, 64, 256, PP | MSB, 1 << 20)
    assert rc == AEC_OK and len(dec) == 1 << 20
    assert hashlib.sha256(dec).hexdigest().startswith("e6e1bf684916")
    rc, enc, _, offs, bits = oracle_encode(dec, 16, 64, 256, PP | MSB)
    assert rc == AEC_OK and enc == typical_rz
    assert used == bits and (bits + 7) // 8 == len(typical_rz)
    # BASELINE config 1: re-encode at block 16 / rsi 128 (observed with the reference: 740174 B)
    rc, enc2, *_ = oracle_encode(dec, 16, 16, 128, PP | MSB)
    assert len(enc2) == 740174 and hashlib.sha256(enc2).hexdigest().startswith("60f1f251f7e6")
    rc, dec2, _ = oracle_decode(enc2, 16, 16, 128, PP | MSB, 1 << 20)
    assert dec2 == dec


def test_golden_vectors(golden):
    for i in range(len(golden)):
        name, bps, bs, rsi, flags, data, expect, declen = golden.case(i)
        rc, enc, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK and enc == expect, name
        rc, dec, _ = oracle_decode(expect, bps, bs, rsi, flags, declen)
        assert rc == AEC_OK and len(dec) == declen, name
        nb = bytes_per_sample(bps, flags)
        n = data.size // nb
        mask = (1 << bps) - 1
        assert np.array_equal(unpack_samples(dec, bps, flags)[:n] & mask,
                              unpack_samples(data, bps, flags)[:n] & mask), name


def test_option_ids(golden):
    """check_code_options.c:25-31: the first ID bits of the stream name the forced option."""
    seen = set()
    for i in range(len(golden)):
        name, bps, bs, rsi, flags, data, expect, _ = golden.case(i)
        if not name.startswith("opt-"):
            continue
        kind = name.split("-")[1]
        idl = id_len_of(bps, flags)
        rc, enc, trace, *_ = oracle_encode(data, bps, bs, rsi, flags, want_trace=True)
        first = enc[0]
        if kind == "zero":
            assert first >> (8 - (idl + 1)) == 0 and trace["option"][0] == OPT_ZERO, name
        elif kind == "se":
            assert first >> (8 - (idl + 1)) == 1 and trace["option"][0] == OPT_SE, name
        elif kind == "uncomp":
            assert first >> (8 - idl) == (1 << idl) - 1 and trace["option"][0] == OPT_UNCOMP, name
        elif kind == "fs":
            assert first >> (8 - idl) == 1 and trace["option"][0] == OPT_SPLIT, name
        else:
            k = int(kind[5:])
            assert first >> (8 - idl) == k + 1 and trace["k"][0] == k, name
        seen.add(kind if not kind.startswith("split") else "split")
    assert seen == {"zero", "se", "uncomp", "fs", "split"}


def test_k_carry(golden):
    """the chosen k depends on the previous block's k, also across an RSI."""
    rc, enc, tr, *_ = oracle_encode(np.array([2] * 8, np.uint8), 8, 8, 1, 0, want_trace=True)
    assert enc == bytes([0x24, 0x92, 0x49, 0x20]) and tr["k"][0] == 0
    for rsi in (1, 2):
        rc, enc, tr, *_ = oracle_encode(np.array([12] * 8 + [2] * 8, np.uint8), 8, 8, rsi, 0,
                                        want_trace=True)
        assert list(tr["k"]) == [3, 2]


def test_padding_and_sizes():
    # empty input -> one zero byte (encode.c:686-695)
    rc, enc, *_ = oracle_encode(b"", 16, 16, 128, PP)
    assert rc == AEC_OK and enc == b"\x00"
    # 11 samples in -> one padded block of 16 out (encode.c:676-684)
    data = pack_samples(np.arange(11) * 3, 16, PP)
    rc, enc, *_ = oracle_encode(data, 16, 16, 128, PP)
    rc, dec, _ = oracle_decode(enc, 16, 16, 128, PP, 64)
    assert len(dec) == 32 and dec[:22] == data.tobytes()
    assert unpack_samples(dec, 16, PP)[11:].tolist() == [30] * 5
    # output too small: STREAM_ERROR (encode.c:944-945), truncated bytes are a prefix
    big = pack_samples(np.arange(4096) * 7 % 65536, 16, PP)
    rc0, full, *_ = oracle_encode(big, 16, 16, 128, PP)
    rc, part, *_ = oracle_encode(big, 16, 16, 128, PP, out_cap=100)
    assert rc == AEC_STREAM_ERROR and part == full[:100]
    # 0 < avail_out < bytes_per_sample at exit -> MEM_ERROR (decode.c:821-823)
    rc, dec, _ = oracle_decode(full, 16, 16, 128, PP, 4096 * 2 - 1)
    assert rc == AEC_MEM_ERROR


def test_conf_errors():
    d = np.zeros(64, np.uint8)
    assert oracle_encode(d, 0, 16, 128, 0)[0] == AEC_CONF_ERROR
    assert oracle_encode(d, 33, 16, 128, 0)[0] == AEC_CONF_ERROR
    assert oracle_encode(d, 8, 12, 128, 0)[0] == AEC_CONF_ERROR
    assert oracle_encode(d, 8, 12, 128, AEC_NOT_ENFORCE)[0] == AEC_OK
    assert oracle_encode(d, 8, 13, 128, AEC_NOT_ENFORCE)[0] == AEC_CONF_ERROR
    assert oracle_encode(d, 8, 16, 4097, 0)[0] == AEC_CONF_ERROR
    assert oracle_encode(d, 8, 8, 128, AEC_RESTRICTED)[0] == AEC_CONF_ERROR   # BASELINE config 5 note
    assert oracle_encode(d, 4, 8, 128, AEC_RESTRICTED)[0] == AEC_OK


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (reference tree absent)")
def test_differential_vs_reference():
    rng = np.random.default_rng(77)
    for it in range(400):
        bps = int(rng.integers(1, 33))
        flags = 0
        if rng.random() < 0.7:
            flags |= AEC_DATA_PREPROCESS
        if rng.random() < 0.5:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.4 and bps > 1:
            flags |= AEC_DATA_SIGNED
        if rng.random() < 0.3:
            flags |= AEC_DATA_3BYTE
        if bps <= 4 and rng.random() < 0.5:
            flags |= AEC_RESTRICTED
        if rng.random() < 0.2:
            flags |= AEC_NOT_ENFORCE
            bs = int(rng.integers(1, 33)) * 2
        else:
            bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 128, 130, 300]))
        n = int(rng.integers(0, 3000))
        if rng.random() < 0.5:
            vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 1, 5, 50, 1000])),
                                       zero_frac=float(rng.choice([0.05, 0.5])))
        else:
            lo = -(1 << (bps - 1)) if flags & AEC_DATA_SIGNED else 0
            hi = (1 << (bps - 1)) - 1 if flags & AEC_DATA_SIGNED else (1 << bps) - 1
            vals = rng.integers(lo, hi + 1, size=n)
        data = pack_samples(vals, bps, flags)
        rc_r, enc_r = ref_encode(data, bps, bs, rsi, flags)
        rc_o, enc_o, *_ = oracle_encode(data, bps, bs, rsi, flags)
        assert (rc_r, enc_r) == (rc_o, enc_o), (it, bps, bs, rsi, flags, n)
        nb = bytes_per_sample(bps, flags)
        cap = ((n + bs - 1) // bs) * bs * nb
        for c in (cap, cap + 40, max(0, cap - int(rng.integers(0, 60)))):
            rc_r, dec_r = ref_decode(enc_r, bps, bs, rsi, flags, c)
            rc_o, dec_o, _ = oracle_decode(enc_r, bps, bs, rsi, flags, c)
            assert rc_r == rc_o and (rc_r != AEC_OK or dec_r == dec_o), (it, bps, bs, rsi, flags, n, c)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (reference tree absent)")
def test_truncated_streams_vs_reference():
    """A stream cut anywhere (also inside a coded data set): the reference's resumable readers release
    every sample whose bits arrived (decode.c:342-400, 423-460, 560-587, 646-657); the oracle must
    release exactly the same ones.  This pins the behaviour the GPU tests hold the product to."""
    rng = np.random.default_rng(78)
    for it in range(150):
        bps = int(rng.choice([8, 12, 16, 24, 32]))
        flags = AEC_DATA_PREPROCESS if rng.random() < 0.8 else 0
        if rng.random() < 0.5:
            flags |= AEC_DATA_MSB
        if rng.random() < 0.4:
            flags |= AEC_DATA_SIGNED
        bs = int(rng.choice([8, 16, 32, 64]))
        rsi = int(rng.choice([1, 3, 16, 64, 130]))
        n = int(rng.integers(bs, 4000))
        vals = random_walk_samples(rng, n, bps, flags, scale=float(rng.choice([0.3, 3, 50, 5000])),
                                   zero_frac=float(rng.choice([0.05, 0.5])))
        data = pack_samples(vals, bps, flags)
        rc, enc = ref_encode(data, bps, bs, rsi, flags)
        assert rc == AEC_OK
        cap = ((n + bs - 1) // bs) * bs * bytes_per_sample(bps, flags)
        for cut in [int(v) for v in rng.integers(0, len(enc) + 1, 6)]:
            rc_r, dec_r = ref_decode(enc[:cut], bps, bs, rsi, flags, cap)
            rc_o, dec_o, _ = oracle_decode(enc[:cut], bps, bs, rsi, flags, cap)
            assert rc_r == rc_o and dec_r == dec_o, (it, bps, bs, rsi, flags, n, cut, len(dec_r), len(dec_o))


def test_pad_rsi_decode():
    """AEC_PAD_RSI (decoder side, reference decode.c:407-408): every RSI starts on a byte boundary.
    The reference encoder never pads (ENABLE_RSI_PADDING is dead code, encode.c:499-505), so such a
    stream is built by concatenating independently coded RSIs."""
    rng = np.random.default_rng(3)
    bps, bs, rsi, flags = 16, 16, 8, AEC_DATA_PREPROCESS
    vals = random_walk_samples(rng, bs * rsi * 5, bps, flags, scale=3.0, zero_frac=0.2)
    data = pack_samples(vals, bps, flags)
    rb = bs * rsi * 2
    stream = b"".join(oracle_encode(data[i:i + rb], bps, bs, rsi, flags)[1] for i in range(0, data.size, rb))
    rc, dec, _ = oracle_decode(stream, bps, bs, rsi, flags | 32, data.size)
    assert rc == AEC_OK and dec == data.tobytes()
    if have_ref():
        rc, dec = ref_decode(stream, bps, bs, rsi, flags | 32, data.size)
        assert rc == AEC_OK and dec == data.tobytes()


@pytest.mark.parametrize("bps,bs,rsi,n_rsi,long_hi", [(16, 16, 8, 40, 120), (8, 8, 128, 4, 200), (32, 32, 16, 8, 400)])
def test_overlong_coded_data_sets_oracle_vs_reference(bps, bs, rsi, n_rsi, long_hi):
    """Streams with coded data sets far longer than any the reference ENCODER writes (split option k = 0 for
    large residuals) are valid input for its decoder (decode.c:462-502); the oracle must agree on them, since
    the GPU tests use it as the checker for exactly these streams."""
    from helpers import craft_overlong_stream
    rng = np.random.default_rng(bps + bs)
    enc = craft_overlong_stream(rng, bps, bs, rsi, n_rsi, {8: 3, 16: 4, 32: 5}[bps], 0.1, long_hi)
    nbytes = n_rsi * rsi * bs * bytes_per_sample(bps, AEC_DATA_PREPROCESS)
    rc_o, dec_o, _ = oracle_decode(enc, bps, bs, rsi, AEC_DATA_PREPROCESS, nbytes)
    assert rc_o == AEC_OK and len(dec_o) == nbytes
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    rc_r, dec_r = ref_decode(enc, bps, bs, rsi, AEC_DATA_PREPROCESS, nbytes)
    assert rc_r == AEC_OK and dec_r == dec_o
