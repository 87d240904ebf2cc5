"""Which scheme the index pass of a bare stream takes (libaec_amd/csrc/aec_idx.hip: launch_index) -- host arithmetic on
the parameters, the size and the average coded RSI (aec_gpu_index_scheme), so it is pinned here without a GPU.  The table
is the dispatch as DESIGN.md section 2 describes it and as tests/bench_short_rsi.py (--edges), tests/bench_index.py and
tests/fuzz_index_gpu.py --time measured it; a change of a threshold shows up here first."""
import pytest

from libaec_amd import gpu

PP, MSB = 8, 4
SERIAL, LOCKED, TABLES, TRUNK, EVERY_BIT, REGIONS = range(6)

CASES = [
    # name, bits per sample, block, rsi, flags, stream bytes, bits per coded RSI, start block, scheme
    ("a 64 KiB chunk with scan lines of 32 pixels", 8, 8, 4, PP, 24_000, 97, 0, EVERY_BIT),
    ("the same, a walk that resumes inside an RSI (round 6: the every-bit scheme's too)", 8, 8, 4, PP, 24_000, 97, 1, EVERY_BIT),
    ("AEC_PAD_RSI, a 64 KiB chunk (round 6)", 8, 8, 4, PP | 32, 24_000, 97, 0, EVERY_BIT),
    ("the 8-bit SZIP shape, a 64 KiB chunk", 8, 8, 128, PP, 23_700, 2960, 0, EVERY_BIT),
    ("the 8-bit SZIP shape, 1 MiB", 8, 8, 128, PP, 379_000, 2960, 0, TABLES),
    ("config 2, 1 GiB: regions (what they do not deliver is left to the window tables)", 16, 16, 128, PP, 190_000_000, 5800, 0, REGIONS),
    ("config 2, 4 MiB: the window tables", 16, 16, 128, PP, 745_000, 5800, 0, TABLES),
    ("config 3, 1 MiB: the tables do not serve it, the trunk's launches would cost 4.5 ms", 32, 32, 4096, PP, 260_000,
     1_040_000, 0, EVERY_BIT),
    ("config 3, 1 GiB", 32, 32, 4096, PP, 260_000_000, 1_040_000, 0, TRUNK),
    ("config 3, 4 GiB: a lane per RSI", 32, 32, 4096, PP, 1_037_000_000, 1_013_000, 0, REGIONS),
    ("the sample file's shape, 1 MiB", 16, 64, 256, PP | MSB, 737_000, 184_000, 0, EVERY_BIT),
    ("the sample file's shape, 1 GiB: entries by plausibility", 16, 64, 256, PP | MSB, 755_000_000, 184_000, 0, LOCKED),
    ("16 MiB of 8-bit data, rsi 32", 8, 8, 32, PP, 5_955_000, 762, 0, LOCKED),
    ("16 MiB of 8-bit data, rsi 33: the tables resolve RSIs that short badly (72 ms)", 8, 8, 33, PP, 5_953_000, 786, 0, LOCKED),
    ("16 MiB of 8-bit data, rsi 48", 8, 8, 48, PP, 5_940_000, 1140, 0, TABLES),
    ("config 2, 256 MiB: below half a gigabit of stream the window tables are through first", 16, 16, 128, PP, 47_700_000, 5800, 0, TABLES),
    ("16 MiB of 16-bit data, rsi 32", 16, 16, 32, PP, 2_930_000, 1470, 0, LOCKED),
    ("4 MiB of the sample file's shape: up to 4 MiB of stream every bit parsed, piece by piece (round 6)", 16, 64, 256, PP | MSB, 2_950_000, 184_000, 0, EVERY_BIT),
    ("8 MiB of the sample file's shape: entries by plausibility", 16, 64, 256, PP | MSB, 5_900_000, 184_000, 0, LOCKED),
    ("16 MiB without the preprocessor: piece by piece", 16, 16, 16, 0, 16_640_000, 4160, 0, EVERY_BIT),
    ("16 MiB of 8-bit data without the preprocessor, rsi 128: the trunk is faster", 8, 8, 128, 0, 16_060_000, 8600, 0, TRUNK),
    ("1 MiB with rsi 1 and long coded data sets", 16, 64, 1, PP, 790_000, 800, 0, EVERY_BIT),
    ("16 MiB with rsi 1 and long coded data sets: scoring chains that carry the count", 24, 64, 1, PP, 7_800_000, 900, 0, LOCKED),
    ("64 MiB of 8-bit noise: RSIs of uncompressed blocks mark no RSI start -- past the schemes that look for reference samples "
     "(round 6); runs of uncompressed coded data sets start chains of their own in the window tables: 11 ms against the "
     "trunk's 25", 8, 8, 128, PP, 70_178_074, 8566, 0, TABLES),
    ("the same stream with a hint that is a look-ahead, not a mean (1.5 means: the ABI's): regions", 8, 8, 128, PP, 70_178_074, 12849, 0, REGIONS),
    ("64 MiB of 16-bit noise", 16, 16, 128, PP, 68_154_718, 33278, 0, TRUNK),
    ("nothing to index", 16, 16, 128, PP, 0, 0, 0, SERIAL),
]


@pytest.mark.parametrize("name,bps,bs,rsi,flags,nbytes,hint,start_block,scheme", CASES, ids=[c[0] for c in CASES])
def test_index_scheme(name, bps, bs, rsi, flags, nbytes, hint, start_block, scheme):
    assert len(gpu.INDEX_SCHEMES) == 6
    got = gpu.index_scheme(bps, bs, rsi, flags, nbytes, hint, start_block)
    assert got == scheme, (name, gpu.INDEX_SCHEMES[got], gpu.INDEX_SCHEMES[scheme])
