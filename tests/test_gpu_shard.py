"""BASELINE config 4's logic on the real kernels, on ONE device: an input is cut into 2, 3 and 8
RSI-aligned shards of unequal size, every shard is planned (aec_gpu_encode_plan_async), the plan
records are "all-gathered" (concatenated: with N ranks that is one 24-byte RCCL all-gather), every
shard is emitted at its global bit offset with the carried k -- both through the host-side carry
(shard.carry_in + aec_gpu_encode_emit_async) and through the device-side one
(aec_gpu_encode_emit_planned_async, no host round trip) -- and the slices are stitched by the
device kernel (aec_gpu_stitch_async).  The result must be the oracle's stream of the whole input
and the single-device stream.  Shapes: config 2/4 (16-bit, block 16, rsi 128) and config 3 (32-bit
signed MSB, block 32, rsi 4096), with a short last RSI and shards that start in the middle of a byte."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import (AEC_DATA_MSB, AEC_DATA_PREPROCESS, AEC_DATA_SIGNED, AEC_OK, ROOT, bytes_per_sample,
                     oracle_encode)

pytestmark = pytest.mark.gpu

PP, MSB, SGN = AEC_DATA_PREPROCESS, AEC_DATA_MSB, AEC_DATA_SIGNED


def gen(kind, nbytes, shard=0):
    lib = C.CDLL(f"{ROOT}/libaec_amd/lib/libaec_datagen.so")
    a = np.empty(nbytes, dtype=np.uint8)
    bps = {0: 2, 1: 4, 2: 1}[kind]
    lib.aec_gen_fill_parallel(C.c_uint(kind), C.c_uint64(shard), C.c_void_p(a.ctypes.data),
                              C.c_size_t(nbytes // bps), C.c_uint(8))
    return a


def cuts(rng, n_rsi, world):
    """world contiguous shards of whole RSIs, unequal, none empty"""
    inner = np.sort(rng.choice(np.arange(1, n_rsi), size=world - 1, replace=False))
    return [0] + [int(v) for v in inner] + [n_rsi]


@pytest.mark.parametrize("kind,bps,bs,rsi,flags,nbytes", [
    (0, 16, 16, 128, PP, (6 << 20) + 4096 * 3 + 16 * 2 * 5 + 6),          # config 2/4 shape, short last RSI
    (1, 32, 32, 4096, PP | MSB | SGN, (16 << 20) + 32 * 4 * 700 + 8),     # config 3 shape, short last RSI
    (2, 8, 8, 128, PP, (1 << 20) + 999),                                   # config 5 kernel shape
])
def test_sharded_stream_equals_single_stream(kind, bps, bs, rsi, flags, nbytes):
    import torch
    from libaec_amd import gpu, shard
    nb = bytes_per_sample(bps, flags)
    data = gen(kind, nbytes // nb * nb + (nbytes % nb))
    rc, want, _, _, total_bits = oracle_encode(data, bps, bs, rsi, flags)
    assert rc == AEC_OK
    rsi_bytes = rsi * bs * nb
    n_rsi = (data.size // nb * nb + rsi_bytes - 1) // rsi_bytes
    d_all = torch.from_numpy(data).cuda()
    single = gpu.Codec(bps, bs, rsi, flags)
    d_ref, n_ref, tb_ref, k_ref, _ = single.encode(d_all)
    assert tb_ref == total_bits and d_ref[:n_ref].cpu().numpy().tobytes() == want
    rng = np.random.default_rng(bps * 7 + rsi)
    for world in (2, 3, 8):
        edges = cuts(rng, n_rsi, world)
        parts = []
        for r in range(world):
            lo, hi = edges[r] * rsi_bytes, min(edges[r + 1] * rsi_bytes, data.size)
            # (clone: every shard starts on an aligned buffer of its own, as on a rank of its own)
            parts.append(d_all[lo:hi].clone())
        codecs = [gpu.Codec(bps, bs, rsi, flags) for _ in range(world)]
        eres = [torch.zeros(24, dtype=torch.uint8, device="cuda") for _ in range(world)]
        for r in range(world):
            codecs[r].encode_plan_async(parts[r], parts[r].numel(), eres[r])
        d_plans = torch.cat(eres)                                   # what the 24-byte all-gather leaves on every rank
        plans = [tuple(int(v) for v in (p["total_bits"], p["k_lo"], p["k_hi"]))
                 for p in d_plans.cpu().numpy().view(gpu.ENC_RESULT_DTYPE)]
        assert sum(b for b, _, _ in plans) == total_bits
        slot = shard.slot_bytes(plans)
        starts_mid_byte = 0
        for planned in (False, True):
            outs = []
            for r in range(world):
                start, k_in = shard.carry_in(plans, r)
                starts_mid_byte += 1 if start % 8 else 0
                d_out = torch.zeros(max(codecs[r].encode_bound(parts[r].numel()), slot), dtype=torch.uint8, device="cuda")
                if planned:
                    if r:                                           # the plan of this context must be the current one
                        codecs[r].encode_plan_async(parts[r], parts[r].numel(), eres[r])
                    codecs[r].encode_emit_planned_async(parts[r], parts[r].numel(), d_out, None, eres[r], d_plans, r)
                else:
                    codecs[r].encode_emit_async(parts[r], parts[r].numel(), d_out, None, eres[r], start % 8, k_in)
                res = eres[r].cpu().numpy().view(gpu.ENC_RESULT_DTYPE)[0]
                assert not res["overflow"] and int(res["total_bits"]) == plans[r][0]
                outs.append(d_out)
            # k after the last shard == k after the single stream
            assert int(eres[-1].cpu().numpy().view(gpu.ENC_RESULT_DTYPE)[0]["k_out"]) == k_ref
            d_gathered = torch.zeros(world * slot + 16, dtype=torch.uint8, device="cuda")
            for r in range(world):
                d_gathered[r * slot:(r + 1) * slot] = outs[r][:slot]
            d_stream = torch.full((len(want) + 64,), 0xAA, dtype=torch.uint8, device="cuda")
            d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
            gpu.stitch_async(d_gathered, slot, d_plans, world, d_stream, d_total)
            assert int(d_total.item()) == len(want)
            got = d_stream[:len(want)].cpu().numpy().tobytes()
            assert got == want, (world, planned, next(i for i in range(len(want)) if got[i] != want[i]))
            # the torch stitch used by the gloo tests gives the same
            st2, n2 = shard.stitch(d_gathered[:world * slot], slot, plans)
            assert n2 == len(want) and st2[:n2].cpu().numpy().tobytes() == want
        if world == 8:
            assert starts_mid_byte > 0          # the case that needs the OR of shared bytes was exercised


def test_device_shard_step_single_rank():
    """DeviceShard (what bench.py times with N ranks) with world = 1: plan, planned emit, stitch."""
    import torch
    from libaec_amd import gpu, shard
    bps, bs, rsi, flags = 16, 16, 128, PP
    data = gen(0, 4 << 20)
    rc, want, *_ = oracle_encode(data, bps, bs, rsi, flags)
    codec = gpu.Codec(bps, bs, rsi, flags)
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.zeros(codec.encode_bound(data.size), dtype=torch.uint8, device="cuda")
    d_eres = torch.zeros(24, dtype=torch.uint8, device="cuda")
    slot = (len(want) + 4096 + 4095) // 4096 * 4096
    sh = shard.DeviceShard(codec, 0, 1, slot)
    d_stream = torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    sh.step(d_in, data.size, d_out, None, d_eres)
    sh.gather_and_stitch(d_out, d_stream, d_total)
    assert int(d_total.item()) == len(want) and d_stream[:len(want)].cpu().numpy().tobytes() == want


def test_c_recipe_over_rccl_world_1():
    """INTEGRATION.md section 5 as a C program (tests/c/shard_rccl.c): plan, ncclAllGather of the 24-byte plan
    records, emit at the global offset, ncclAllGather of the slices, stitch -- against librccl with a communicator of
    one rank; the stream must be aec_buffer_encode's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "shard_rccl")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(root, "libaec_amd", "lib")
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c", "-O1", os.path.join(root, "tests", "c", "shard_rccl.c"),
                    "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    "-L", lib, "-l:libaec.so.0", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
                    "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "shard_rccl ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_path_with_ranks_sharing_one_gpu(world):
    """BASELINE config 4's path end to end with N > 1 PROCESSES: plan, exchange of the 24-byte records, emit at the
    global bit offset, all-gather of the slices, stitch, decode of every rank's shard out of the stitched stream --
    bench.py under torch.distributed.run with all ranks on cuda:0 and gloo (through the host) in place of RCCL,
    which needs a GPU per rank.  bench.py asserts the round trips itself and, on rank 0, that the stitched stream is
    byte for byte what ONE encoder makes of the inputs of all ranks one behind the other."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", str(world), "--share-gpu", "--size-mib", "256", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["value"] > 0
    assert "one bit-exact stream" in out["config"]["parallelism"]
    # the line proves its own N: what the process group says, every rank's device, the size of
    # the all-gather, and a checksum of the stitched stream that all ranks agreed on
    comm = out["comm"]
    assert comm["backend"] == "gloo" and comm["world_size"] == world and len(comm["devices"]) == world
    assert sorted(d["rank"] for d in comm["devices"]) == list(range(world))
    assert len({d["pid"] for d in comm["devices"]}) == world and comm["distinct_devices"] == 1      # (--share-gpu)
    assert comm["allgather_bytes"] > 0 and comm["plan_exchange_bytes"] == 24 * world
    assert comm["stitched"]["bytes"] > 0 and "sha16" in comm["stitched"]


def test_bench_starts_its_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT torchrun (the form of the driver's N = 1 command with another N): bench.py
    must start the two ranks itself and rank 0's line must say n_gpus == 2.  (--share-gpu: the test box has one GPU.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--size-mib", "256",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "one bit-exact stream" in out["config"]["parallelism"]
