"""N>1 path on CPU: env sharding and the episode-return gather over `gloo`, world_size 2
(and a ragged 3-rank case).  The data path itself needs no collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from finrl_amd.distributed import (gather_episode_returns, reduce_return_stats, shard_env_kwargs,
                                   shard_range)


def test_shard_ranges_partition_the_batch():
    for E, R in [(65536 * 8, 8), (10, 3), (7, 8), (524288, 8), (5, 1)]:
        seen = []
        for r in range(R):
            lo, hi = shard_range(E, r, R)
            seen += list(range(lo, hi))
        assert seen == list(range(E))
    cash = np.arange(10, dtype=np.float64)
    sh = np.arange(30).reshape(10, 3)
    n, kw = shard_env_kwargs(10, 1, 3, initial_amount=cash, num_stock_shares=sh, hmax=5)
    assert n == 3 and kw["hmax"] == 5
    np.testing.assert_array_equal(kw["initial_amount"], cash[4:7])
    np.testing.assert_array_equal(kw["num_stock_shares"], sh[4:7])
    n, kw = shard_env_kwargs(10, 0, 2, initial_amount=1e6, num_stock_shares=[1, 2, 3])
    assert n == 5 and kw["initial_amount"] == 1e6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_envs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(global_envs, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32) * 0.5 + 1.0     # value encodes global id
    out = gather_episode_returns(local, global_envs)
    st = reduce_return_stats(local)
    q.put((rank, out.numpy().copy(), st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,global_envs", [(2, 256), (2, 7), (3, 10)])
def test_gather_episode_returns_gloo(world, global_envs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, global_envs, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(global_envs, dtype=np.float32) * 0.5 + 1.0
    for rank, out, st in res:
        np.testing.assert_array_equal(out, expect)
        assert st["count"] == global_envs
        assert st["min"] == expect.min() and st["max"] == expect.max()
        assert abs(st["mean"] - expect.mean()) < 1e-9
