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


def _worker_missing_global(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(7, rank, world)                 # ragged shards: 4 + 3
    local = torch.arange(lo, hi, dtype=torch.float32)
    res = []
    try:
        gather_episode_returns(local)                    # size of the collective unknowable locally
        res.append("no error")
    except ValueError as ex:
        res.append(str(ex))
    try:                                                  # a shard of the wrong size raises before
        gather_episode_returns(local[:1], 7)              # the collective (on every rank here)
        res.append("no error")
    except ValueError as ex:
        res.append(str(ex))
    out = gather_episode_returns(local, 7)                # and the ranks are still in step
    q.put((rank, res, out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_requires_global_envs_when_sharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_global, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msgs, out in res:
        assert "global_envs is required" in msgs[0] and "holds 1 envs" in msgs[1]
        np.testing.assert_array_equal(out, np.arange(7, dtype=np.float32))


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


# ---------------------------------------------------------------------------------------------
# bench.py's timed region over gloo: the collective runs inside the timed region at every episode
# end -- and once at its end when no episode finished there -- and the JSON `rccl` block reports
# how many ranks took part.  A CPU stand-in implements the Workload protocol (the envs themselves
# have no CPU path).
# ---------------------------------------------------------------------------------------------
class _StubWork:
    def __init__(self, lo, hi, episode_len):
        self.lo, self.hi, self.episode_len = lo, hi, episode_len
        self.n_steps = self.n_resets = 0

    def step(self, i):
        self.n_steps += 1

    def reset(self):
        self.n_resets += 1

    def episode_return(self):
        return torch.arange(self.lo, self.hi, dtype=torch.float32) + 0.25 * self.n_steps


class _Ev:
    def record(self):
        self.t = 0.0

    def elapsed_time(self, other):
        return 1.0


def _bench_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for steps, warmup, prewarm, L in [(3, 2, 0, 5), (2, 2, 0, 5), (12, 1, 7, 5), (4, 0, 0, None)]:
        lo, hi = shard_range(64, rank, world)
        w = _StubWork(lo, hi, L)
        wall, dev_ms, rccl = bench.timed_region(w, steps, warmup, prewarm, world, dist,
                                                lambda: None, lambda: (_Ev(), _Ev()),
                                                global_envs=64)
        res.append((steps, warmup, prewarm, L, w.n_steps, w.n_resets, rccl))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_gathers_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        (a, b, c, d) = out[rank]
        # episode of 5 steps ends exactly at the last timed step: one gather, at the episode end
        assert a[4] == 5 and a[6] == dict(world=2, backend="gloo", gathers=1, gathered=64)
        # no episode end inside the timed region: one gather at its end all the same
        assert b[6]["gathers"] == 1 and b[6]["gathered"] == 64
        # prewarm (7 + 64 launches) is followed by a reset and does not count; 12 timed steps after
        # 1 warm-up step cross two episode ends
        assert c[4] == 7 + 64 + 1 + 12 and c[5] == 1 and c[6]["gathers"] == 2
        # envs without a common episode length: the single end-of-region gather
        assert d[6]["gathers"] == 1


def test_every_env_kind_shards():
    from finrl_amd.distributed import env_class
    for kind, name in [("stock", "VecStockTradingEnv"), ("stocknp", "VecStockTradingEnvNP"),
                       ("portfolio", "VecStockPortfolioEnv"), ("crypto", "VecCryptoEnv"),
                       ("cashpenalty", "VecCashPenaltyEnv"), ("stoploss", "VecStopLossEnv")]:
        cls = env_class(kind)
        assert cls.__name__ == name and hasattr(cls, "episode_return") and hasattr(cls, "step")
    with pytest.raises(ValueError):
        env_class("nope")
    n, kw = shard_env_kwargs(10, 2, 3, initial_capital=np.arange(10.0), gamma=0.9)
    assert n == 3 and kw["gamma"] == 0.9
    np.testing.assert_array_equal(kw["initial_capital"], np.arange(10.0)[7:10])


# ---------------------------------------------------------------------------------------------
# `python bench.py --gpus N` outside torchrun starts its own ranks (the shape of the driver's
# single-GPU command with N > 1): a child `python -m torch.distributed.run ... bench.py`, never an
# exec, before torch is imported.  `--env launcher-selftest` swaps the kernels for a counter and
# RCCL for gloo, so the launch / rendezvous / gather / JSON plumbing runs where no GPU exists.
# ---------------------------------------------------------------------------------------------
def _run_bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd="/tmp",
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_self_launches_its_ranks():
    r, out = _run_bench(["--gpus", "2", "--env", "launcher-selftest", "--steps", "6", "--warmup", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None and out["n_gpus"] == 2 and out["steps"] == 6
    assert out["rccl"] == dict(world=2, backend="gloo", gathers=1, gathered=2 * out["config"]["envs_per_gpu"])
    assert out["config"]["global_envs"] == 2 * out["config"]["envs_per_gpu"]
    assert out["stub_steps_run"] == 8            # warm-up + timed steps, once per rank
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1     # rank 0 only


def test_bench_single_rank_and_world_mismatch():
    r, out = _run_bench(["--gpus", "1", "--env", "launcher-selftest", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0 and out["n_gpus"] == 1 and out["rccl"] is None
    # inside a (here: one-rank) torchrun environment the ranks are the launcher's business:
    # a --gpus that disagrees with WORLD_SIZE is an error message, not an assert
    r, out = _run_bench(["--gpus", "2", "--env", "launcher-selftest", "--steps", "3", "--warmup", "1"],
                        env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and out is None and "WORLD_SIZE=1" in r.stderr
