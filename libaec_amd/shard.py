"""One adaptive-entropy stream coded on several devices (mode ii).

The coder's only state that crosses a block boundary is the bit position and the carried k.
Each rank therefore (1) PLANs its contiguous, RSI-aligned shard (total bits and the k clamp
(k_lo, k_hi) of the shard -- aec_gpu_encode_plan_async), (2) learns the plans of the ranks before
it through one tiny all-gather, (3) EMITs its shard at bit `start % 8` of its own buffer with the
composed carry-in k (aec_gpu_encode_emit_async), and (4) the byte slices are reassembled on every
rank by ONE all-gather (RCCL over xGMI when the group's backend is "nccl") followed by a local
stitch that ORs the shared boundary bytes.  The result is byte-identical to coding the whole input
on one device (and to the CPU reference).

The host-side helpers below (exchange_plans / carry_in / stitch) are backend-agnostic torch code:
the CPU tests drive them with gloo and a test encoder.  On the device the same step runs WITHOUT a
host round trip (DeviceShard): the 24-byte plan records are all-gathered as they lie in HBM,
aec_gpu_encode_emit_planned_async derives start bit and carried k from them in a kernel, and
aec_gpu_stitch_async compacts the gathered slices (libaec_amd/csrc/aec_shard.hip); a C caller does
the same with two ncclAllGather calls.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_ranges(total_bytes: int, world: int, rsi_bytes: int) -> List[Tuple[int, int]]:
    """Contiguous shards, every one but the last a whole number of RSIs."""
    n_rsi = (total_bytes + rsi_bytes - 1) // rsi_bytes
    out, start = [], 0
    for r in range(world):
        first = (n_rsi * r) // world
        last = (n_rsi * (r + 1)) // world
        lo = min(first * rsi_bytes, total_bytes)
        hi = min(last * rsi_bytes, total_bytes) if r + 1 < world else total_bytes
        out.append((lo, hi - lo))
        start = hi
    return out


def clamp_k(k: int, lo: int, hi: int) -> int:
    return min(max(k, lo), hi)


def carry_in(plans: Sequence[Tuple[int, int, int]], rank: int, k0: int = 0) -> Tuple[int, int]:
    """plans[r] = (total_bits, k_lo, k_hi) of rank r.  Returns (absolute start bit, k_in) of `rank`:
    the bit lengths add up and the k clamps compose in rank order (aec_lane.h kclamp_then)."""
    start, k = 0, k0
    for bits, lo, hi in plans[:rank]:
        start += bits
        k = clamp_k(k, lo, hi)
    return start, k


def exchange_plans(bits: int, k_lo: int, k_hi: int, group=None, device="cpu") -> List[Tuple[int, int, int]]:
    """all-gather of the three plan numbers of every rank (24 bytes per rank)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([bits, k_lo, k_hi], dtype=torch.int64, device=device)
    got = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    return [tuple(int(v) for v in t.tolist()) for t in got]


def slot_bytes(plans: Sequence[Tuple[int, int, int]], align: int = 4096) -> int:
    """Size of one all-gather slot: the largest slice (its bits plus up to 7 leading bits)."""
    need = max((b + 7 + 7) // 8 for b, _, _ in plans) if plans else 1
    return (need + align - 1) // align * align


def gather_slices(local: torch.Tensor, slot: int, group=None) -> torch.Tensor:
    """ONE all-gather of the per-rank slices (each padded to `slot` bytes) -> [world * slot]."""
    world = dist.get_world_size(group)
    assert local.dtype == torch.uint8 and local.numel() >= slot
    out = torch.empty(world * slot, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local[:slot].contiguous(), group=group)
    return out


def stitch(gathered: torch.Tensor, slot: int, plans: Sequence[Tuple[int, int, int]],
           out: torch.Tensor = None) -> Tuple[torch.Tensor, int]:
    """Compact the gathered slices into one stream.  Slice r starts at bit (start_r % 8) of its
    slot; its first byte may share bits with the last byte of slice r-1: those are OR-ed.
    Returns (stream tensor, length in bytes: max(1, ceil(total_bits / 8)))."""
    total_bits = sum(b for b, _, _ in plans)
    nbytes = max(1, (total_bits + 7) // 8)
    if out is None:
        out = torch.zeros(nbytes + 8, dtype=torch.uint8, device=gathered.device)
    else:
        out[: nbytes + 1].zero_()
    start = 0
    for r, (bits, _, _) in enumerate(plans):
        lead = start % 8
        n = (lead + bits + 7) // 8
        if n:
            dst = start // 8
            src = gathered[r * slot: r * slot + n]
            first = out[dst].clone()
            out[dst: dst + n] = src
            out[dst] |= first          # boundary byte shared with the previous slice
        start += bits
    return out, nbytes


def _all_gather_bytes(out: torch.Tensor, local: torch.Tensor, group=None) -> None:
    """out[r * n : (r + 1) * n] = rank r's `local` (n bytes, device tensors).  RCCL gathers them where they lie; any
    other backend (gloo: ranks that share ONE GPU in a test) goes through host copies."""
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return
    world = dist.get_world_size(group)
    mine = local.detach().cpu().contiguous()
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    out.copy_(torch.cat(got).to(out.device))


class DeviceShard:
    """One rank's part of a stream coded on `world` devices, all on the device (no host sync per step).

        sh = DeviceShard(codec, rank, world, slot, group)
        sh.step(d_in, nbytes, d_out, d_off, d_eres)     # plan -> all-gather 24 B -> emit at the global offset
        sh.gather_and_stitch(d_out, d_stream, d_total)  # all-gather of the slices -> one stream (device kernel)

    `slot` (bytes, a multiple of 4096) must hold the largest slice; it has to be known on the host to
    size the all-gather, e.g. from a first run (slot_bytes(plans)) or from a bound on the ratio.
    world == 1 works without a process group (the gathers are copies)."""

    def __init__(self, codec, rank, world, slot, group=None):
        from . import gpu
        self.gpu, self.codec, self.rank, self.world, self.slot, self.group = gpu, codec, rank, world, slot, group
        dev = torch.device("cuda", torch.cuda.current_device())
        self.d_plans = torch.zeros(world * 24, dtype=torch.uint8, device=dev)
        self.d_gathered = torch.zeros(world * slot + 16, dtype=torch.uint8, device=dev)

    def step(self, d_in, nbytes, d_out, d_off, d_eres):
        self.codec.encode_plan_async(d_in, nbytes, d_eres)
        if self.world > 1:
            _all_gather_bytes(self.d_plans, d_eres[:24], self.group)
        else:
            self.d_plans.copy_(d_eres[:24])
        self.codec.encode_emit_planned_async(d_in, nbytes, d_out, d_off, d_eres, self.d_plans, self.rank)

    def gather_and_stitch(self, d_out, d_stream, d_total=None):
        view = self.d_gathered[: self.world * self.slot]
        if self.world > 1:
            _all_gather_bytes(view, d_out[: self.slot], self.group)
        else:
            view.copy_(d_out[: self.slot])
        self.gpu.stitch_async(self.d_gathered, self.slot, self.d_plans, self.world, d_stream, d_total)
