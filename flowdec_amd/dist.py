"""Batch sharding of enhance() across the GPUs of one node.

Every clip is independent end to end (per-clip normalisation, per-sample GroupNorm, shared deterministic time
embedding, convolution algorithm chosen by IMAGE size only), so the only communication is moving waveforms: rank r
processes clips [lo, hi) of the global batch and the results are all-gathered (ONE `all_gather_into_tensor` per call).
No collective sits on the data path of the solver itself.  Works with the `nccl` (= RCCL over xGMI) backend on GPUs and
with `gloo` (CPU tensors, or GPU tensors staged through the host) in the tests.

    out = sharded_enhance(model, y, N=6, solver="euler", seed=1234)     # every rank gets all B waveforms

N-GPU == 1-GPU, bit for bit: the initial noise of clip i (the reference draws it inside enhance from the device RNG,
flowdec/model.py:512,530-536) depends on the GLOBAL clip index i only -- either sliced from a caller-provided global
`noise` tensor, or drawn from the per-clip stream (seed, i) -- never on the rank layout.
"""
from typing import Callable, List, Optional, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def _world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _needs_host_staging(group=None) -> bool:
    """True when the group's collectives cannot take device tensors (gloo).  Composite backend strings such as
    'cpu:gloo,cuda:nccl' carry a device backend: only a plain 'gloo' group has none."""
    import torch.distributed as dist
    b = str(dist.get_backend(group)).lower()
    return "nccl" not in b and "cuda:" not in b


def all_gather_shards(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """Rank r holds items shard_range(n_items, r, world) of a [n_items, ...] tensor -> the whole tensor on every rank.
    One `all_gather_into_tensor` into a [world, max_shard, ...] buffer; with an even split the result is a view of that
    buffer (no further copy), with an uneven one the padding rows are dropped by one index_select.  Runs the collective
    even at world size 1 when a process group exists (that is how the RCCL path is exercised on a 1-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("all_gather_shards needs an initialised process group (torch.distributed.init_process_group)")
    world, rank = _world(group)
    sizes = shard_sizes(n_items, world)
    assert local.shape[0] == sizes[rank], f"rank {rank}: local shard has {local.shape[0]} items, expected {sizes[rank]}"
    pad, tail = max(sizes), tuple(local.shape[1:])
    if local.shape[0] == pad:
        send = local.contiguous()
    else:
        send = local.new_zeros((pad,) + tail)
        send[: local.shape[0]] = local
    if send.is_cuda and _needs_host_staging(group):   # gloo moves host memory: stage device tensors through the host
        out = send.new_empty((world * pad,) + tail, device="cpu")
        dist.all_gather_into_tensor(out, send.cpu(), group=group)
        out = out.to(send.device)
    else:
        out = local.new_empty((world * pad,) + tail)
        dist.all_gather_into_tensor(out, send, group=group)
    if all(s == pad for s in sizes):
        return out
    keep = torch.cat([torch.arange(r * pad, r * pad + s) for r, s in enumerate(sizes)]).to(out.device)
    return out.index_select(0, keep)


def sharded_apply(fn: Callable[[torch.Tensor], torch.Tensor], y: torch.Tensor, group=None, always_gather: bool = False) -> torch.Tensor:
    """Apply `fn` (e.g. `lambda yb: model.enhance(yb, N=6, noise=...)`) to this rank's slice of the batch dimension of `y`
    and return the full result on every rank.  `fn` must map [b, ...] -> [b, ...] with the same trailing shape.
    Without a process group (or with one rank, unless `always_gather`) this is just `fn(y)`."""
    world, rank = _world(group)
    if world == 1 and not always_gather:
        return fn(y)
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("sharded_apply(always_gather=True) needs an initialised process group")
    lo, hi = shard_range(y.shape[0], rank, world)
    # (an idle rank -- fewer clips than ranks -- still takes part in the collective; `fn` keeps the trailing shape, dtype and device)
    local = fn(y[lo:hi]) if hi > lo else y.new_zeros((0,) + tuple(y.shape[1:]))
    return all_gather_shards(local, y.shape[0], group)


def clip_noise(seed: int, index: int, shape, device) -> torch.Tensor:
    """The initial noise of GLOBAL clip `index`: complex64 standard normal of `shape` from its own Philox stream
    (seed, index) -- the same values whatever the batch, the shard or the rank the clip is processed in."""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1000003 + int(index)) & 0x7FFFFFFFFFFFFFFF)
    return torch.randn(shape, dtype=torch.complex64, device=device, generator=g)


def sharded_enhance(model, y: torch.Tensor, N: int = 50, solver: str = "euler", noise: Optional[torch.Tensor] = None,
                    seed: Optional[int] = None, generator: Optional[torch.Generator] = None, group=None, always_gather: bool = False,
                    stats: Optional[dict] = None, **enhance_kwargs) -> torch.Tensor:
    """`model.enhance(y, N=N, solver=solver)` for a global batch y [B, 1, L] sharded by clip over the ranks of `group`;
    every rank returns all B enhanced waveforms [B, 1, L] on y's device.  Result == the single-process call, bit for bit:

      noise=      global initial noise [B, 1, F, T_pad] complex64 (every rank passes the same tensor or at least its own
                  rows): rank r uses rows [lo, hi);
      seed=       clip i draws from the stream (seed, i) (`clip_noise`); the default when nothing is given: rank 0 draws a
                  seed and broadcasts it;
      generator=  one generator seeded IDENTICALLY on every rank: the full [B, ...] noise is drawn and sliced, which equals
                  `model.enhance(y, generator=g)` of a single process (costs B x 1.5 MB per second of audio of device memory).

    y may live on the host (pinned or not): only this rank's rows are copied to the model's device, and the gathered result
    is copied back -- the "H2D of waveform -> D2H of waveform" path of SURVEY 8(d).  `stats`, if given, receives
    {"local_s", "gather_s"} host-clock seconds (it synchronises the device, use it for measurements only)."""
    import time
    if y.ndim != 3 or y.shape[1] != 1:
        raise RuntimeError(f"sharded_enhance expects a batch [B, 1, L] (got {tuple(y.shape)})")
    world, rank = _world(group)
    B, Lw = y.shape[0], y.shape[-1]
    lo, hi = shard_range(B, rank, world)
    dev = model.device
    out_device = y.device
    t0 = time.perf_counter() if stats is not None else 0.0
    local = None
    if hi > lo:
        yl = y[lo:hi].to(dev, non_blocking=True)
        if noise is not None:
            nz = noise[lo:hi]
        else:
            cfg = model.feature_extractor._cfg()
            n_freq, T = cfg["n_fft"] // 2 + 1, 1 + Lw // cfg["hop"]      # fd_num_frames / fd_padded_frames (pad_spec: multiple of 64)
            Tp = 64 * ((T + 63) // 64)
            if generator is not None:
                nz = _full_draw(generator, B, n_freq, Tp, dev)[lo:hi]
            else:
                if seed is None:
                    seed = _shared_seed(dev, group)
                nz = torch.stack([clip_noise(seed, i, (1, n_freq, Tp), dev) for i in range(lo, hi)])
        local = model.enhance(yl, N=N, solver=solver, noise=nz, **enhance_kwargs)
    elif noise is None and generator is not None:
        # idle rank (fewer clips than ranks): draw and discard the full noise, so that the generators of all ranks stay identical for
        # the next call -- the contract of `generator=`
        cfg = model.feature_extractor._cfg()
        T = 1 + Lw // cfg["hop"]
        _full_draw(generator, B, cfg["n_fft"] // 2 + 1, 64 * ((T + 63) // 64), dev)
    elif noise is None and seed is None:
        _shared_seed(dev, group)   # idle rank: still takes part in the seed broadcast
    if stats is not None:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        stats["local_s"] = stats.get("local_s", 0.0) + (t1 - t0)
    if world == 1 and not always_gather:
        out = local
    else:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("sharded_enhance(always_gather=True) needs an initialised process group")
        if local is None:
            local = torch.empty((0, 1, Lw), dtype=torch.float32, device=dev)
        out = all_gather_shards(local, B, group)
    if out.device != out_device:
        out = out.to(out_device)
    if stats is not None:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        stats["gather_s"] = stats.get("gather_s", 0.0) + (time.perf_counter() - t1)
    return out


def _full_draw(generator: torch.Generator, B: int, n_freq: int, Tp: int, dev) -> torch.Tensor:
    """The [B, 1, F, T_pad] complex64 noise of the whole global batch from `generator` (on the generator's device)."""
    return torch.randn((B, 1, n_freq, Tp), dtype=torch.complex64, device=generator.device, generator=generator).to(dev)


def _shared_seed(dev, group=None) -> int:
    """A fresh seed that every rank agrees on (rank 0 draws, one 8-byte broadcast)."""
    import torch.distributed as dist
    world, rank = _world(group)
    s = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if world > 1:
        s = s if _needs_host_staging(group) else s.to(dev)
        dist.broadcast(s, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(s.item())
