"""Env-instance data parallelism over the GPUs of one node.

Environments are fully independent (no cross-env term anywhere in step/reset; the market
panel is read-only and replicated per GPU), so rank r of R owns the contiguous env range
[r*E/R, (r+1)*E/R) and the data path needs NO collective.  The only exchange is the gather
of per-env episode returns at episode end: one small all-gather (256 KB/rank at 65,536 envs
per GPU) over RCCL/xGMI -- latency-bound, amortised over the T steps of an episode.

One process per GPU (torchrun-style); backend "nccl" is RCCL on ROCm, "gloo" is used by the
CPU tests of this logic.
"""
from __future__ import annotations

import numpy as np


def shard_range(global_envs: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of the global env index owned by `rank` (sizes differ by at
    most one when world does not divide global_envs)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(global_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_env_kwargs(global_envs: int, rank: int, world: int, **kw):
    """Slice per-env constructor arguments (initial_amount [E], num_stock_shares [E, N]) down
    to this rank's envs; scalars / per-ticker vectors pass through unchanged."""
    lo, hi = shard_range(global_envs, rank, world)
    out = dict(kw)
    ia = kw.get("initial_amount")
    if ia is not None and np.ndim(ia) == 1 and len(ia) == global_envs:
        out["initial_amount"] = np.asarray(ia)[lo:hi]
    ns = kw.get("num_stock_shares")
    if ns is not None and np.ndim(ns) == 2 and len(ns) == global_envs:
        out["num_stock_shares"] = np.asarray(ns)[lo:hi]
    return hi - lo, out


def make_sharded_env(panel, global_envs: int, *, rank=None, world=None, device=None, **kw):
    """This rank's VecStockTradingEnv shard (panel replicated on the local GPU)."""
    import torch
    import torch.distributed as dist
    from .vec_env import VecStockTradingEnv
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    n_local, kw = shard_env_kwargs(global_envs, rank, world, **kw)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return VecStockTradingEnv(panel, n_local, device=device, **kw)


def gather_episode_returns(local_returns, global_envs: int = None, group=None):
    """All-gather the per-env episode returns of every rank into global env order.

    local_returns: 1-D tensor (this rank's envs, in local order).  Returns a 1-D tensor of
    length sum(shard sizes) on the same device, identical on every rank.  Equal shard sizes
    use one `all_gather_into_tensor` (a single RCCL all-gather); ragged shards fall back to
    the list form.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_returns.clone()
    world = dist.get_world_size(group)
    n = local_returns.numel()
    if global_envs is None or global_envs == n * world:
        out = torch.empty(n * world, dtype=local_returns.dtype, device=local_returns.device)
        try:
            dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group)
            return out
        except (RuntimeError, NotImplementedError):
            pass
    sizes = [shard_range(global_envs if global_envs is not None else n * world, r, world)
             for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(nmax, dtype=local_returns.dtype, device=local_returns.device)
    pad[:n] = local_returns
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)])


def reduce_return_stats(local_returns, group=None):
    """(count, mean, min, max) of episode returns over all ranks via all-reduce on a tiny
    vector -- the alternative to gathering when only summary statistics are needed."""
    import torch
    import torch.distributed as dist
    x = local_returns.to(torch.float64)
    v = torch.stack([torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device),
                     x.sum()])
    mn, mx = x.min().clone(), x.max().clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return dict(count=int(v[0].item()), mean=float((v[1] / v[0]).item()),
                min=float(mn.item()), max=float(mx.item()))
