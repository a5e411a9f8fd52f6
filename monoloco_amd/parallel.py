"""Multi-GPU layout of the path: rows (persons) are independent in eval mode, so a batch is cut
into contiguous row shards, one per rank (one process per GPU), weights are replicated, there is
no communication during compute and exactly ONE collective at the end: a gather of the (rows, 5)
fp32 (x, y, z, d, sigma) block to rank 0 (RCCL over xGMI with the ``nccl`` backend; ``gloo`` in
the CPU tests).  xGMI is point-to-point, so the direct all-to-one gather (7 concurrent receives
on rank 0, one per link) is link-optimal; nothing is ringed.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total_rows, world_size, rank):
    """Contiguous, balanced [lo, hi) row range of `rank` (first `total % world` ranks get one more)."""
    base, rem = divmod(int(total_rows), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_device_index(local_rank):
    """The HIP device of this process under torchrun: device `local_rank` of the node's GPUs (every rank sees them all); a rank beyond
    the visible devices has none of its own and must not silently share another rank's."""
    have = torch.cuda.device_count()
    if local_rank >= have:
        raise RuntimeError("local rank %d has no HIP device of its own: torch.cuda.device_count() = %d" % (local_rank, have))
    return local_rank


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world_size, local_rank).  A single process is (0, 1, 0) with no group -- unless ``force``: then a
    world-size-1 group is created too, so that the very code path of N > 1 (process group, collectives on the real
    backend) can be exercised on a one-GPU box (bench.py --force-distributed, tests/test_gpu_parallel.py)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            dev_index = local_device_index(local)
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class RowGather:
    """Pre-allocated gather of per-rank (rows_r, width) blocks to rank `dst`, in rank order.

    ``mode='gather'`` (default): direct all-to-one, only rank `dst` holds the result.  ``mode='all_gather'``: one
    ``all_gather_into_tensor`` on equal shards -- every rank ends up with the full block (7x the traffic, but a
    single fused RCCL kernel; bench.py --gather all_gather times it against the default)."""

    def __init__(self, total_rows, width, device, dtype=torch.float32, dst=0, group=None, mode='gather'):
        assert mode in ('gather', 'all_gather')
        self.group = group
        self.dst = dst
        self.mode = mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bounds = [shard_bounds(total_rows, self.world, r) for r in range(self.world)]
        self.full = None
        self.parts = None
        equal = len({hi - lo for lo, hi in self.bounds}) == 1
        self._equal = equal
        self._all = mode == 'all_gather' and equal and dist.is_initialized()   # ragged shards fall back to the p2p gather
        if self.rank == dst or self._all:
            self.full = torch.empty((total_rows, width), dtype=dtype, device=device)
            self.parts = [self.full[lo:hi] for lo, hi in self.bounds]

    def __call__(self, local_block):
        """Collective: returns the (total_rows, width) tensor on rank dst, None elsewhere."""
        if self.world == 1 and not dist.is_initialized():
            self.full.copy_(local_block)
            return self.full
        if self._all:
            dist.all_gather_into_tensor(self.full, local_block.contiguous(), group=self.group)
        elif self._equal:
            dist.gather(local_block, gather_list=self.parts if self.rank == self.dst else None, dst=self.dst,
                        group=self.group)
        else:  # ragged shards: point-to-point (gather needs equal sizes)
            if self.rank == self.dst:
                reqs = []
                for r, part in enumerate(self.parts):
                    if r == self.dst:
                        part.copy_(local_block)
                    elif part.numel():
                        reqs.append(dist.irecv(part, src=r, group=self.group))
                for q in reqs:
                    q.wait()
            elif local_block.numel():
                dist.send(local_block, dst=self.dst, group=self.group)
        return self.full


class ShardedRows:
    """The whole N > 1 layout in one object: rows [0, total) cut into contiguous per-rank shards (``lo``, ``hi``), a
    caller-supplied function computes the local (hi - lo, width) block -- no communication -- and ONE gather brings
    the blocks to rank ``dst`` in row order.  bench.py and the gloo tests drive the same class."""

    def __init__(self, total_rows, width, device, dtype=torch.float32, dst=0, group=None, mode='gather'):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = shard_bounds(total_rows, self.world, self.rank)
        self.gather = RowGather(total_rows, width, device, dtype=dtype, dst=dst, group=group, mode=mode)

    @property
    def local_rows(self):
        return self.hi - self.lo

    def run(self, fn):
        """fn(lo, hi) -> (hi - lo, width) tensor on the gather's device; returns the gathered rows on rank dst."""
        block = fn(self.lo, self.hi)
        assert block.shape[0] == self.hi - self.lo, "the local block must cover exactly this rank's shard"
        return self.gather(block)
