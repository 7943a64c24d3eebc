"""Multi-GPU batch split.

Proofs are independent and share only the read-only commitment key, so the path shards by a
contiguous split of the batch: one process per GPU, B/G proofs each, NO data-path collective.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) is used
only for the barrier around the timed region, one broadcast of the commitment key, and for reducing the
results: max of the elapsed time, sum (and per-rank list) of the accepted-proof counts, optional gather of the
per-proof flags to rank 0.
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


def reduce_result(dist, elapsed: float, accepted: int, device, gather: bool = False):
    """(max over ranks of elapsed seconds, sum over ranks of accepted proofs[, accepted proofs of every rank])."""
    import torch

    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    ac = torch.tensor([accepted], dtype=torch.int64, device=device)
    per_rank = [accepted]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if gather:
            bufs = [torch.empty_like(ac) for _ in range(dist.get_world_size())]
            dist.all_gather(bufs, ac)
            per_rank = [int(b.item()) for b in bufs]
        dist.all_reduce(ac, op=dist.ReduceOp.SUM)
    if gather:
        return float(el.item()), int(ac.item()), per_rank
    return float(el.item()), int(ac.item())


def broadcast_key(dist, A, device=None, src: int = 0):
    """The commitment key is the only data the ranks share: `src` holds the dense [a1;a2] slab
    ((n+l) x k polynomials, 48 KiB at (1,3,1) N=1024, 4.25 MiB at config 5) and broadcasts it once; every rank then
    loads it with rzk_key_load(_dev).  A: int64 tensor of the key's shape on every rank (contents only matter on src).
    device: where the collective runs (the tensor's own device for nccl / RCCL; "cpu" for the gloo rehearsal)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return A
    if device is None or A.device == device:
        dist.broadcast(A, src=src)
        return A
    staged = A.to(device)
    dist.broadcast(staged, src=src)
    return staged.to(A.device)


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
