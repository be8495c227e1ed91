"""One process per GPU: shapes are independent units, sharded contiguously by rank.

The reference shards a batch with single-process torch.nn.DataParallel
(/root/reference/train_multigpu.py:136-140) and hand-slices the per-shape ground-truth
meshes by torch.cuda.current_device() (/root/reference/parallel.py:162-171).  Here every
rank owns its shapes for the whole step; the static grid topology is replicated; the only
exchange on the hot path is an all-gather of the per-shape loss scalars (RCCL over xGMI on
GPUs — backend "nccl" IS RCCL on ROCm —, gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [start, stop) of `n_items` shapes owned by `rank`; earlier ranks take the
    remainder, so sizes differ by at most one (ragged batches are allowed)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items: int, world: int):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def all_gather_losses(local_losses: torch.Tensor, n_items: int | None = None, group=None) -> torch.Tensor:
    """local_losses [n_local, k] (or [n_local]) -> [n_items, k] in global shape order on every
    rank.  Equal shard sizes use one all_gather_into_tensor; ragged shards pad to the largest
    shard and trim."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_losses
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x = local_losses.contiguous()
    n_local = x.shape[0]
    if n_items is None:
        n_items = n_local * world
    sizes = shard_sizes(n_items, world)
    if sizes[rank] != n_local:
        raise ValueError("rank %d holds %d shapes, expected %d" % (rank, n_local, sizes[rank]))
    if len(set(sizes)) == 1:
        out = x.new_empty((world * n_local,) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    m = max(sizes)
    pad = x.new_zeros((m,) + tuple(x.shape[1:]))
    pad[:n_local] = x
    out = x.new_empty((world * m,) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)], 0)


class LossGather:
    """The all-gather of the per-shape losses, taken OFF the critical path: `submit` enqueues the
    collective with async_op=True (RCCL runs it on its own stream once the producer kernels have
    finished) and returns the PREVIOUS step's gathered losses, so the next step's kernels never wait
    for the exchange; `flush` waits for the outstanding one.  Equal shard sizes only (the bench /
    training batch layout); ragged shards use all_gather_losses."""

    def __init__(self, group=None):
        self.group = group
        self._work = None
        self._out = None

    def submit(self, local_losses: torch.Tensor):
        prev = self.flush()
        if not (dist.is_available() and dist.is_initialized()):
            self._out = local_losses
            return prev
        x = local_losses.contiguous()
        world = dist.get_world_size(self.group)
        self._out = x.new_empty((world * x.shape[0],) + tuple(x.shape[1:]))
        self._keep = x                                   # keep the input alive until the collective has run
        self._work = dist.all_gather_into_tensor(self._out, x, group=self.group, async_op=True)
        return prev

    def flush(self):
        """Wait for the outstanding collective (if any) and return its result (None if none)."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        out, self._out = self._out, None
        return out
