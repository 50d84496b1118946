"""Batch sharding of enhance() across the GPUs of one node.

Every clip is independent end to end (per-clip normalisation, per-sample GroupNorm, shared deterministic time
embedding), so the only communication is moving waveforms: rank r processes clips [lo, hi) of the batch and the
results are all-gathered.  No collective sits on the data path of the solver itself.  Works with the `nccl` (= RCCL
over xGMI) backend on GPUs and with `gloo` on CPU tensors (used by the tests).
"""
from typing import Callable, List, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def sharded_apply(fn: Callable[[torch.Tensor], torch.Tensor], y: torch.Tensor, group=None) -> torch.Tensor:
    """Apply `fn` (e.g. `lambda yb: model.enhance(yb, N=6)`) to this rank's slice of the batch dimension of `y` and
    return the full result on every rank.  `fn` must map [b, ...] -> [b, ...] with the same trailing shape."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return fn(y)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(y.shape[0], rank, world)
    local = fn(y[lo:hi]) if hi > lo else y.new_zeros((0,) + tuple(y.shape[1:]))
    sizes = shard_sizes(y.shape[0], world)
    pad = max(sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: hi - lo] = local
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return torch.cat([g[:n] for g, n in zip(gathered, sizes)], dim=0)
