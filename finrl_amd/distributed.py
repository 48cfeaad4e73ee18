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


# constructor arguments that may be given per env ([E] / [E, N]) and are then sliced per rank
_PER_ENV_KWARGS = ("initial_amount", "num_stock_shares", "initial_capital", "initial_stocks")


def shard_env_kwargs(global_envs: int, rank: int, world: int, **kw):
    """Slice per-env constructor arguments (e.g. initial_amount [E], num_stock_shares [E, N]) down
    to this rank's envs; scalars / per-ticker vectors pass through unchanged."""
    lo, hi = shard_range(global_envs, rank, world)
    out = dict(kw)
    for name in _PER_ENV_KWARGS:
        v = kw.get(name)
        if v is not None and np.ndim(v) >= 1 and len(v) == global_envs and \
                (np.ndim(v) == 2 or name in ("initial_amount", "initial_capital")):
            out[name] = np.asarray(v)[lo:hi]
    return hi - lo, out


def env_class(kind):
    """Batched env class by name: every env of SURVEY.md 8(a)/(f-3) shards the same way."""
    if kind == "stock":
        from .vec_env import VecStockTradingEnv as cls
    elif kind == "stocknp":
        from .vec_stocknp import VecStockTradingEnvNP as cls
    elif kind == "portfolio":
        from .vec_portfolio import VecStockPortfolioEnv as cls
    elif kind == "crypto":
        from .vec_crypto import VecCryptoEnv as cls
    elif kind == "cashpenalty":
        from .vec_cashpenalty import VecCashPenaltyEnv as cls
    elif kind == "stoploss":
        from .vec_cashpenalty import VecStopLossEnv as cls
    else:
        raise ValueError(f"unknown env kind {kind!r}")
    return cls


def make_sharded_env(panel, global_envs: int, *, kind="stock", rank=None, world=None,
                     device=None, **kw):
    """This rank's shard of a global batch of `kind` envs (market panel / config replicated on the
    local GPU).  `panel` is whatever the env class takes first: a StockPanel / PortfolioPanel /
    CashPenaltyPanel, or the reference-style config dict of the array-state and crypto envs."""
    import torch
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    n_local, kw = shard_env_kwargs(global_envs, rank, world, **kw)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return env_class(kind)(panel, n_local, device=device, **kw)


class _PendingGather:
    """Handle of an asynchronous gather: ``wait()`` blocks the current stream on the collective and
    returns the gathered tensor."""

    def __init__(self, out, work, finish=None):
        self._out, self._work, self._finish = out, work, finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish(self._out) if self._finish is not None else self._out


def gather_episode_returns(local_returns, global_envs: int = None, group=None, async_op=False):
    """All-gather the per-env episode returns of every rank into global env order.

    async_op=True: the collective is only enqueued (on RCCL's own stream) and a handle is returned;
    env steps launched meanwhile overlap with it, ``handle.wait()`` yields the gathered tensor.

    local_returns: 1-D tensor (this rank's envs, in local order).  Returns a 1-D tensor of
    length global_envs on the same device, identical on every rank.  With more than one rank
    ``global_envs`` is REQUIRED: the shape of the collective is derived from it alone, by
    arithmetic every rank evaluates identically (never from the local tensor's size, and never by
    catching an error on one rank -- either would leave the ranks in different collectives):
    equal shards (global_envs divisible by the world size) use one `all_gather_into_tensor` (a
    single RCCL all-gather; gloo implements it too); ragged shards pad to the largest shard and use
    the list form.  A rank whose tensor does not have shard_range(global_envs, rank, world) entries
    raises before entering the collective.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = local_returns.clone()
        return _PendingGather(out, None) if async_op else out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = local_returns.numel()
    if global_envs is None:
        raise ValueError("gather_episode_returns: global_envs is required when world > 1 "
                         "(ragged shards cannot be told from equal ones locally)")
    lo, hi = shard_range(global_envs, rank, world)
    if n != hi - lo:
        raise ValueError(f"rank {rank} holds {n} envs, its shard of {global_envs} has {hi - lo}")
    if global_envs % world == 0:
        out = torch.empty(n * world, dtype=local_returns.dtype, device=local_returns.device)
        work = dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group,
                                           async_op=async_op)
        return _PendingGather(out, work) if async_op else out
    sizes = [shard_range(global_envs, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(nmax, dtype=local_returns.dtype, device=local_returns.device)
    pad[:n] = local_returns
    parts = [torch.empty_like(pad) for _ in range(world)]
    work = dist.all_gather(parts, pad, group=group, async_op=async_op)

    def finish(ps):
        return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(ps, sizes)])
    return _PendingGather(parts, work, finish) if async_op else finish(parts)


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
