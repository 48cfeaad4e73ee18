"""The N > 1 path on real kernels: two ranks (both on the box's single GPU, `gloo` for the
collective -- RCCL needs one GPU per rank) each build their shard of a global batch with
make_sharded_env, collect PPO rollouts into device buffers, run the GAE scan and gather the
episode returns.  Envs are independent, so the gathered result must equal the single-process run of
the whole batch bit for bit (BASELINE configs[3] and [4] shapes in miniature)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(kind):
    rng = np.random.default_rng(42)
    if kind == "crypto":
        T, N, W, E = 40, 10, 40, 140
        price = 10.0 ** rng.uniform(0, 4, N) * np.exp(np.cumsum(rng.normal(0, 0.003, (T, N)), 0))
        cfg = {"price_array": price, "tech_array": rng.normal(0, 3000, (T, W))}
        return cfg, dict(initial_capital=2e5), E, N, 1 + N + W
    T, N, K, E = 30, 100, 2, 140                     # NASDAQ-100 shape, turbulence threshold
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    risk = np.abs(rng.normal(0, 30, T))
    cash = 1e6 * rng.uniform(0.5, 1.5, E)            # per-env argument: sliced per rank
    return (close, tech, risk), dict(initial_amount=cash, turbulence_threshold=45.0), E, N, \
        1 + 2 * N + K * N


def _run(kind, rank, world, n_seg, n_steps):
    from finrl_amd import StockPanel
    from finrl_amd.distributed import gather_episode_returns, make_sharded_env, shard_range
    from finrl_amd.rollout import RolloutBuffer
    data, kw, E, N, D = _inputs(kind)
    panel = StockPanel(*data) if kind == "stock" else data
    env = make_sharded_env(panel, E, kind=kind, rank=rank, world=world, device="cuda:0", **kw)
    lo, hi = shard_range(E, rank, world)
    assert env.num_envs == hi - lo
    g = torch.Generator().manual_seed(7)
    acts = torch.rand(n_seg * n_steps, E, N, generator=g) * 2 - 1     # global action streams
    vals = torch.rand(n_seg * n_steps + 1, E, generator=g)
    buf = RolloutBuffer(n_steps, hi - lo, D, N, device="cuda:0")
    obs = env.reset()
    adv = []
    for s in range(n_seg):
        def policy(o, _s=s, _t=[0]):
            t = _s * n_steps + _t[0]
            _t[0] += 1
            return acts[t, lo:hi].cuda(), vals[t, lo:hi].cuda(), torch.zeros(hi - lo, device="cuda:0")
        obs = buf.collect(env, policy, obs)
        a, r = buf.compute_returns_and_advantage(vals[(s + 1) * n_steps, lo:hi].cuda())
        adv.append(torch.stack([a, r]).cpu())
    ret = gather_episode_returns(env.episode_return(), E)
    return ret.cpu().numpy(), torch.stack(adv).numpy(), obs.cpu().numpy(), buf.dones.cpu().numpy()


def _worker(kind, rank, world, port, n_seg, n_steps, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _run(kind, rank, world, n_seg, n_steps)
    q.put((rank,) + out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_seg,n_steps", [("crypto", 5, 16), ("stock", 4, 16)])
def test_two_rank_shards_equal_the_single_batch(kind, n_seg, n_steps):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    import torch.multiprocessing as mp
    from finrl_amd.distributed import shard_range
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(kind, r, 2, port, n_seg, n_steps, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ret1, adv1, obs1, dones1 = _run(kind, 0, 1, n_seg, n_steps)       # the whole batch, one process
    E = ret1.shape[0]
    assert dones1.any(), "the run must cross an episode end"
    for rank in (0, 1):
        ret, adv, obs, dones = res[rank]
        lo, hi = shard_range(E, rank, 2)
        np.testing.assert_array_equal(ret, ret1)                      # gathered: global order
        np.testing.assert_array_equal(adv, adv1[:, :, :, lo:hi])      # advantages / returns
        np.testing.assert_array_equal(obs, obs1[lo:hi])
        np.testing.assert_array_equal(dones, dones1[:, lo:hi])
