"""Multi-GPU batch split.

Proofs are independent and share only the read-only commitment key, so the path shards by a
contiguous split of the batch: one process per GPU, B/G proofs each, NO data-path collective.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) is used
only for the barrier around the timed region and for reducing the results: max of the elapsed
time, sum of the accepted-proof counts, optional gather of the per-proof flags to rank 0.
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a batch of `total` proofs owned by `rank`; sizes differ by <= 1."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank / world / total")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier(dist, sync_device=None) -> None:
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if sync_device is not None:
        import torch

        torch.cuda.synchronize(sync_device)


def reduce_result(dist, elapsed: float, accepted: int, device) -> Tuple[float, int]:
    """(max over ranks of elapsed seconds, sum over ranks of accepted proofs)."""
    import torch

    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    ac = torch.tensor([accepted], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ac, op=dist.ReduceOp.SUM)
    return float(el.item()), int(ac.item())


def gather_flags(dist, flags, total: int, dst: int = 0):
    """Per-proof uint8 flags of every rank's shard -> the full [total] vector on `dst` (None elsewhere)."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return flags
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=flags.dtype, device=flags.device)
    pad[: flags.numel()] = flags
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
