"""The N>1 path on CPU: world_size-2 gloo processes exercise the batch split and the result reduction
bench.py uses on the GPUs (there with backend "nccl" = RCCL).  No ring arithmetic here — the kernels
need a GPU; what is covered is that shards partition the batch and that the reductions are right."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ring_zk_amd import shard


def test_shard_range_partitions():
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            pieces = [shard.shard_range(total, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == total
            for (a, b), (c, d) in zip(pieces, pieces[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in pieces]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.shard_range(total, rank, world)
        # stand-in for the verify flags of this rank's shard: proof i is "accepted" unless i % 5 == 0
        flags = torch.tensor([(i % 5) != 0 for i in range(lo, hi)], dtype=torch.uint8)
        shard.barrier(dist)
        elapsed = 0.25 * (rank + 1)
        mx, acc = shard.reduce_result(dist, elapsed, int(flags.sum()), torch.device("cpu"))
        mx2, acc2, per_rank = shard.reduce_result(dist, elapsed, int(flags.sum()), torch.device("cpu"), gather=True)
        assert (mx2, acc2) == (mx, acc) and per_rank[rank] == int(flags.sum()) and sum(per_rank) == acc
        # key broadcast: rank 0 owns the [a1;a2] slab, the others start from zeros (bench.py --broadcast-key)
        key = torch.arange(2 * 3 * 16, dtype=torch.int64).reshape(2, 3, 16) - 40 if rank == 0 else torch.zeros(2, 3, 16, dtype=torch.int64)
        key = shard.broadcast_key(dist, key, torch.device("cpu"))
        assert torch.equal(key, torch.arange(2 * 3 * 16, dtype=torch.int64).reshape(2, 3, 16) - 40)
        full = shard.gather_flags(dist, flags, total, dst=0)
        q.put((rank, lo, hi, mx, acc, None if full is None else full.numpy().tolist()))
        shard.barrier(dist)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 4097])
def test_two_rank_split_and_reduction(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_flags = [int(i % 5 != 0) for i in range(total)]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == total
    for rank, lo, hi, mx, acc, full in res:
        assert mx == pytest.approx(0.5)            # max over ranks
        assert acc == sum(expect_flags)            # sum over ranks
        if rank == 0:
            assert full == expect_flags
        else:
            assert full is None
