"""Multi-rank path on CPU: world_size 2 over gloo.  The sharding / carry / gather / stitch logic of
libaec_amd.shard is exercised with a test encoder (tests/emul, which runs the device per-lane code
on the CPU and honours start_bit / k_in like aec_gpu_encode_emit_async); the stitched stream must
be byte-identical to the oracle's single-stream encoding of the whole input."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import AEC_DATA_PREPROCESS, oracle_encode, pack_samples, random_walk_samples


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, data_path, params, out_path):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libaec_amd import shard
    from test_lane_emul import EMUL_SO, emul_encode

    lib = C.CDLL(EMUL_SO)
    lib.emul_encode.restype = C.c_int
    bps, bs, rsi, flags = params
    data = np.load(data_path)
    rsi_bytes = rsi * bs * 2
    lo, n = shard.shard_ranges(data.size, world, rsi_bytes)[rank]
    mine = data[lo:lo + n]
    # PLAN: bits and the k clamp of this shard (k_out for k_in = 0 and k_in = kmax are its ends)
    _, _, bits, k_from0, _, _ = emul_encode(lib, mine, bps, bs, rsi, flags, 0, 0)
    _, _, _, k_from_max, _, _ = emul_encode(lib, mine, bps, bs, rsi, flags, 0, 13)
    plans = shard.exchange_plans(bits, k_from0, k_from_max)
    start, k_in = shard.carry_in(plans, rank)
    # EMIT at the global bit offset
    _, out, bits2, _, _, _ = emul_encode(lib, mine, bps, bs, rsi, flags, start % 8, k_in)
    assert bits2 == bits
    slot = shard.slot_bytes(plans, align=64)
    local = torch.zeros(slot, dtype=torch.uint8)
    nb = (start % 8 + bits + 7) // 8
    local[:nb] = torch.from_numpy(out[:nb].copy())
    gathered = shard.gather_slices(local, slot)
    stream, nbytes = shard.stitch(gathered, slot, plans)
    if rank == 0:
        np.save(out_path, stream[:nbytes].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_single_stream_over_ranks(tmp_path, world):
    from test_lane_emul import EMUL_SO
    if not os.path.exists(EMUL_SO):
        pytest.skip("tests/emul harness not built")
    rng = np.random.default_rng(11)
    bps, bs, rsi, flags = 16, 16, 8, AEC_DATA_PREPROCESS
    n = bs * rsi * 11 + 5                      # shards of unequal size, a short last RSI
    vals = random_walk_samples(rng, n, bps, flags, scale=2.0, zero_frac=0.3)
    data = pack_samples(vals, bps, flags)
    data_path, out_path = str(tmp_path / "in.npy"), str(tmp_path / "out.npy")
    np.save(data_path, data)
    mp.spawn(_worker, args=(world, _free_port(), data_path, (bps, bs, rsi, flags), out_path), nprocs=world, join=True)
    rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
    got = np.load(out_path).tobytes()
    assert got == want


def test_shard_ranges_cover_input():
    from libaec_amd import shard
    for total, world, rb in ((1000, 4, 64), (4096, 8, 4096), (1 << 20, 3, 4096), (10, 2, 64)):
        r = shard.shard_ranges(total, world, rb)
        assert r[0][0] == 0 and sum(n for _, n in r) == total
        for (lo, n), (lo2, _) in zip(r, r[1:]):
            assert lo + n == lo2 and (lo2 % rb == 0 or lo2 == total)
