"""The N > 1 layout on CPU: contiguous row shards, one gather to rank 0 (gloo, world_size 2 and 3)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monoloco_amd import parallel


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 8):
            b = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_bounds(total, world, rank)
    # stand-in for the per-rank device pipeline: a deterministic function of the global row id
    rows = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
    sharded = parallel.ShardedRows(total, 5, torch.device('cpu'))
    assert (sharded.lo, sharded.hi) == (lo, hi)
    for _ in range(2):  # the pre-allocated gather is reusable step after step (this is bench.py's N > 1 step)
        full = sharded.run(lambda a, b: rows)
    if rank == 0:
        ref = torch.arange(total, dtype=torch.float32).unsqueeze(1) * torch.tensor([[1., 2., 3., 4., 5.]])
        q.put(bool(torch.equal(full, ref)))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 4096), (2, 4097), (3, 1000)])
def test_row_gather_gloo(world, total):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True
