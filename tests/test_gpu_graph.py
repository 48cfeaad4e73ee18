"""hipGraph capture of a rollout segment (finrl_amd/graph.py): the env step is a plain launch on
the capturing stream, so a captured segment must reproduce the eager one bit for bit."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


def _crypto(E, seed=0):
    from finrl_amd.vec_crypto import VecCryptoEnv
    rng = np.random.default_rng(seed)
    T, N, W = 400, 10, 40
    price = 10.0 ** rng.uniform(0, 4.5, N) * np.exp(np.cumsum(rng.normal(0, 5e-4, (T, N)), axis=0))
    return VecCryptoEnv({"price_array": price, "tech_array": rng.normal(0, 3000, (T, W))}, E,
                        auto_reset=True), N


def test_graphed_segment_equals_eager_and_is_replayable():
    _need_gpu()
    from finrl_amd.graph import GraphedSegment
    from finrl_amd.rollout import RolloutBuffer
    E, n_steps = 4096, 16
    w = None

    def policy(obs):                       # a tiny capturable "network": tanh(obs @ w)
        a = torch.tanh(obs @ w)
        return a, a.sum(1), a.mean(1)

    env_e, N = _crypto(E)
    env_g, _ = _crypto(E)
    w = torch.randn(env_e.state_dim, N, device="cuda") * 1e-3
    buf_e = RolloutBuffer(n_steps, E, env_e.state_dim, N)
    buf_g = RolloutBuffer(n_steps, E, env_g.state_dim, N)
    obs_e, obs_g = env_e.reset().clone(), env_g.reset().clone()
    seg = GraphedSegment(env_g, policy, buf_g)
    for rep in range(3):                   # three consecutive segments: state carries over
        last_e = buf_e.collect(env_e, policy, obs_e).clone()
        seg.replay(obs_g)
        last_g = buf_g.obs[n_steps].clone()
        for name in ("obs", "actions", "rewards", "dones", "values"):
            assert torch.equal(getattr(buf_e, name), getattr(buf_g, name)), (rep, name)
        for k in env_e.state:
            assert torch.equal(env_e.state[k], env_g.state[k]), (rep, k)
        obs_e, obs_g = last_e, last_g
    # launch-bound regime: one replay call vs 16 x (policy + copies + step) host calls
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        seg.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        buf_e.collect(env_e, policy, buf_e.obs[n_steps])
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 20
    print(f"segment of {n_steps} steps x {E} envs: eager {t_eager * 1e6:.0f} us, graph {t_graph * 1e6:.0f} us")
    assert t_graph < t_eager


def test_stock_env_segment_in_graph_matches_eager():
    """The headline env through the same capture path (its step() takes out= as well)."""
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.graph import GraphedSegment
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_env import VecStockTradingEnv
    close, tech, risk = bench.synth_panel()
    close, tech, risk = close[:40], tech[:40], risk[:40]        # episodes roll over inside a segment
    E, N, n_steps = 2048, 30, 24
    envs = [VecStockTradingEnv(StockPanel(close, tech, risk), E, **bench.ENV_KW) for _ in range(2)]
    w = torch.randn(envs[0].obs.shape[1], N, device="cuda") * 1e-6

    def policy(obs):
        a = torch.tanh(obs @ w)
        return a, a.sum(1), a.mean(1)

    bufs = [RolloutBuffer(n_steps, E, envs[0].obs.shape[1], N) for _ in range(2)]
    o0, o1 = envs[0].reset().clone(), envs[1].reset().clone()
    seg = GraphedSegment(envs[1], policy, bufs[1])
    for rep in range(2):
        last0 = bufs[0].collect(envs[0], policy, o0).clone()
        seg.replay(o1)
        for name in ("obs", "actions", "rewards", "dones"):
            assert torch.equal(getattr(bufs[0], name), getattr(bufs[1], name)), (rep, name)
        for k in envs[0].state:
            assert torch.equal(envs[0].state[k], envs[1].state[k]), (rep, k)
        if rep == 1:
            assert int(bufs[0].dones.sum()) > 0                  # an episode boundary was crossed
        o0, o1 = last0, bufs[1].obs[n_steps].clone()


def test_two_handles_on_two_streams_do_not_interfere():
    """One handle per stream: two envs of different shapes (32-wide and 128-wide
    kernels, dynamic-LDS opt-in included) stepped interleaved on two HIP streams give the same
    results as stepping each alone."""
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    shapes = [(30, 3000), (100, 1000)]
    gens = [torch.Generator(device="cuda") for _ in shapes]
    for g in gens:
        g.manual_seed(3)
    panels = []
    for N, E in shapes:
        c, t, r = bench.synth_panel(N=N)
        panels.append(StockPanel(c[:60], t[:60], r[:60]))
    acts = [[torch.rand(E, N, generator=g, device="cuda") * 2 - 1 for _ in range(70)]
            for (N, E), g in zip(shapes, gens)]
    torch.cuda.synchronize()

    def make():
        return [VecStockTradingEnv(p, E, **bench.ENV_KW) for p, (N, E) in zip(panels, shapes)]

    serial = make()
    outs_serial = []
    for env, a in zip(serial, acts):
        env.reset()
        for x in a:
            env.step(x)
        outs_serial.append({k: v.clone() for k, v in env.state.items()})
    streams = [torch.cuda.Stream() for _ in shapes]
    inter = make()
    for env, s in zip(inter, streams):
        with torch.cuda.stream(s):
            env.reset()
    for t in range(70):
        for env, s, a in zip(inter, streams, acts):
            with torch.cuda.stream(s):
                env.step(a[t])
    torch.cuda.synchronize()
    for env, ref in zip(inter, outs_serial):
        for k in ref:
            assert torch.equal(env.state[k], ref[k]), k


def test_multi_round_step_under_graph_capture():
    """A stock batch with more 64-env groups than one resident round (two launches per step) captured in a
    hipGraph: the replay equals the eager run bit for bit (every launch of a step lands in the capture)."""
    _need_gpu()
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    rng = np.random.default_rng(2)
    T, N, K, E = 12, 30, 2, 70_000
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.02, (T, N)), axis=0))
    panel = StockPanel(close, rng.normal(0, 1, (T, K, N)), np.abs(rng.normal(0, 30, T)))
    a_env = VecStockTradingEnv(panel, E, hmax=30, initial_amount=40_000)
    b_env = VecStockTradingEnv(panel, E, hmax=30, initial_amount=40_000)
    acts = [torch.rand(E, N, device="cuda") * 2 - 1 for _ in range(4)]
    a_env.reset(); b_env.reset()
    b_env.step(acts[0]); a_env.step(acts[0])          # first call outside the capture (occupancy query, caches)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pass
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(1, 4):
            b_env.step(acts[t])
    # the capture did not execute anything: bring b to the state before the captured steps, then replay
    for k, v in a_env.state.items():
        b_env.state[k].copy_(v)
    b_env.obs.copy_(a_env.obs)
    for t in range(1, 4):
        a_env.step(acts[t])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a_env.obs, b_env.obs) and torch.equal(a_env.reward, b_env.reward)
    for k in a_env.state:
        assert torch.equal(a_env.state[k], b_env.state[k]), k
