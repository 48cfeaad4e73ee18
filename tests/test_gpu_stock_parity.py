"""HIP path (through the C ABI) vs (i) the committed reference fixtures and (ii) the CPU
oracle on seeded random batches.  Needs an MI355X: run with `-m gpu` via gpurun.

Bar: holdings / day / done / trades bit-exact; cash, cost and the fp64 reward kept in state
bit-exact (same fp64 operation order as the reference); float32 outputs (obs, reward) equal
to the float32 cast of the oracle's doubles -- i.e. zero tolerance, well inside the 1e-5
relative bound BASELINE.json states.
"""
import ctypes as C

import numpy as np
import pytest

from _golden import StockFixture, stock_fixture_names

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box "
                    "(there is no CPU fallback to hide behind)")


def _make_env(fx_or_panel, E, **kw):
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    if isinstance(fx_or_panel, StockFixture):
        fx = fx_or_panel
        panel = StockPanel(fx.close, fx.tech, fx.risk)
        k = fx.env_kwargs()
        k.update(kw)
        return VecStockTradingEnv(panel, E, **k)
    return VecStockTradingEnv(fx_or_panel, E, **kw)


@pytest.mark.parametrize("name", stock_fixture_names())
def test_hip_matches_reference_fixture(name):
    """Replay the reference-generated fixture on E=70 identical envs (one full wave + a
    6-lane tail) with gym semantics (manual reset after done), compare every step."""
    _need_gpu()
    fx = StockFixture(name)
    z = fx.z
    E = 70
    env = _make_env(fx, E, auto_reset=False)
    env.enable_realised()
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    if -1 in resets:
        obs = env.reset().cpu().numpy()
        np.testing.assert_array_equal(
            obs, np.broadcast_to(resets[-1].astype(np.float32), obs.shape))
    else:
        obs = env.observe().cpu().numpy()
        np.testing.assert_array_equal(obs[0], z["ctor_obs"].astype(np.float32))
    tj = 0
    for s in range(fx.S):
        a = torch.from_numpy(np.broadcast_to(fx.actions[s], (E, fx.N)).copy()).cuda()
        obs, rew, done, _ = env.step(a)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        st = env.state_numpy()
        real = env.realised.cpu().numpy()
        for e in (0, 63, 64, E - 1):
            assert bool(done[e]) == bool(z["done"][s]), (s, e)
            assert st["day"][e] == z["day"][s], (s, e)
            np.testing.assert_array_equal(st["shares"][e], z["shares"][s], err_msg=f"step {s}")
            assert st["trades"][e] == z["trades"][s], (s, e)
            assert st["cash"][e] == z["cash"][s], (s, e, st["cash"][e], z["cash"][s])
            assert st["cost"][e] == z["cost"][s], (s, e)
            assert st["last_reward"][e] == z["reward"][s], (s, e)
            assert st["turbulence"][e] == z["turbulence"][s], (s, e)
            assert rew[e] == np.float32(z["reward"][s]), (s, e)
            np.testing.assert_array_equal(real[e], z["realised"][s], err_msg=f"step {s}")
            if "obs" in z.files:
                np.testing.assert_array_equal(obs[e], z["obs"][s].astype(np.float32),
                                              err_msg=f"obs step {s} env {e}")
        if z["done"][s]:
            stats = env.episode_stats().cpu().numpy()
            am = z[f"asset_memory_{tj}"]
            assert stats[0, 0] == am[0] and stats[E - 1, 0] == am[0]
            assert stats[0, 3] == z["cost"][s] and stats[0, 4] == z["trades"][s]
            sh = fx.sharpe(tj)
            if np.isnan(sh):
                assert np.isnan(stats[0, 5])
            else:
                assert stats[0, 5] == pytest.approx(sh, rel=1e-9, abs=1e-12)
            tj += 1
            obs = env.reset().cpu().numpy()
            np.testing.assert_array_equal(
                obs, np.broadcast_to(resets[s].astype(np.float32), obs.shape))
    assert tj >= 2


def _random_panel(seed, T, N, K, flag_frac=0.03):
    rng = np.random.default_rng(seed)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    if K:
        tech[:, 0, :][rng.random((T, N)) < flag_frac] = 1.0
    risk = np.abs(rng.normal(0, 30, T))
    return close, tech, risk


@pytest.mark.parametrize("cfg", [
    dict(E=1000, T=40, N=30, K=8, steps=100, thr=None, cash=1_000_000, hmax=100),
    dict(E=777, T=25, N=30, K=8, steps=60, thr=45.0, cash=60_000, hmax=100),
    dict(E=130, T=30, N=7, K=3, steps=70, thr=None, cash=5_000, hmax=20),
    dict(E=64, T=12, N=1, K=2, steps=30, thr=30.0, cash=1_000, hmax=10),
    dict(E=257, T=20, N=32, K=1, steps=45, thr=None, cash=100_000, hmax=1000),
    dict(E=65, T=9, N=16, K=0, steps=20, thr=None, cash=30_000, hmax=100),
    # 128-wide kernel variant (NASDAQ-100 shape and its edges)
    dict(E=200, T=14, N=100, K=8, steps=32, thr=40.0, cash=400_000, hmax=100),
    dict(E=70, T=10, N=33, K=1, steps=24, thr=None, cash=50_000, hmax=50),
    dict(E=64, T=9, N=128, K=1, steps=20, thr=None, cash=900_000, hmax=200),
    # N = 100 compile-time variant: int16 key limit (hmax <= 255), a 1-lane tail block, and the
    # generic 128-wide kernel it falls back to above that limit
    dict(E=65, T=9, N=100, K=2, steps=20, thr=None, cash=2_000_000, hmax=255),
    dict(E=130, T=9, N=100, K=1, steps=20, thr=60.0, cash=3_000_000, hmax=1000),
    # 64-wide variant (33..64 tickers)
    dict(E=300, T=16, N=50, K=4, steps=36, thr=35.0, cash=200_000, hmax=100),
    dict(E=129, T=11, N=64, K=2, steps=26, thr=None, cash=80_000, hmax=300),
    dict(E=70, T=10, N=65, K=1, steps=24, thr=None, cash=50_000, hmax=50),
])
def test_hip_matches_oracle_random_batch(cfg):
    """Distinct action streams per env, DummyVecEnv auto-reset semantics, per-env initial
    cash / shares; every output compared with the CPU oracle at every step."""
    _need_gpu()
    from finrl_amd import StockPanel
    from oracle.stock import StockOracle
    E, T, N, K = cfg["E"], cfg["T"], cfg["N"], cfg["K"]
    rng = np.random.default_rng(E + T)
    close, tech, risk = _random_panel(E, T, N, K)
    cash0 = cfg["cash"] * rng.uniform(0.5, 1.5, E)
    sh0 = rng.integers(0, 15, (E, N))
    kw = dict(hmax=cfg["hmax"], initial_amount=cash0, num_stock_shares=sh0,
              buy_cost_pct=0.0013, sell_cost_pct=0.0007, reward_scaling=1e-4,
              turbulence_threshold=cfg["thr"])
    orc = StockOracle(close, tech, risk, n_envs=E, **kw)
    env = _make_env(StockPanel(close, tech, risk), E, auto_reset=True, **kw)
    env.enable_terminal_obs()
    o_obs = orc.reset()
    g_obs = env.reset().cpu().numpy()
    np.testing.assert_array_equal(g_obs, o_obs.astype(np.float32))
    n_done = 0
    for s in range(cfg["steps"]):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        a[rng.random((E, N)) < 0.05] = 0.0
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        g_obs, g_rew, g_done = g_obs.cpu().numpy(), g_rew.cpu().numpy(), g_done.cpu().numpy()
        np.testing.assert_array_equal(g_done.astype(bool), o_done, err_msg=f"done step {s}")
        np.testing.assert_array_equal(g_rew, o_rew.astype(np.float32), err_msg=f"reward step {s}")
        np.testing.assert_array_equal(g_obs, o_obs.astype(np.float32), err_msg=f"obs step {s}")
        st, os_ = env.state_numpy(), orc.state()
        np.testing.assert_array_equal(st["shares"], os_["shares"], err_msg=f"shares step {s}")
        np.testing.assert_array_equal(st["cash"], os_["cash"], err_msg=f"cash step {s}")
        np.testing.assert_array_equal(st["cost"], os_["cost"], err_msg=f"cost step {s}")
        for k in ("day", "price_day", "trades", "episode"):
            np.testing.assert_array_equal(st[k], os_[k], err_msg=f"{k} step {s}")
        np.testing.assert_array_equal(st["last_reward"], os_["last_reward"])
        if o_done.any():
            n_done += 1
            t = env.term_obs.cpu().numpy()
            np.testing.assert_array_equal(t[o_done], o_term[o_done].astype(np.float32))
        gs, os2 = env.episode_stats().cpu().numpy(), orc.episode_stats()
        np.testing.assert_array_equal(gs[:, :5], os2[:, :5], err_msg=f"stats step {s}")
        np.testing.assert_allclose(gs[:, 5], os2[:, 5], rtol=1e-9, atol=1e-12, equal_nan=True)
    assert n_done >= 2


@pytest.mark.parametrize("N,K,hint", [(30, 8, True), (30, 8, False), (7, 3, True), (100, 3, True),
                                      (50, 2, True), (50, 2, False)])
def test_hip_desynchronised_envs(N, K, hint):
    """Envs that are NOT in lock-step (different days inside one wave, via masked resets):
    exercises the per-row reload path of the observation writer and per-lane price gathers, in
    every kernel variant (32-wide, the N = 100 one, 64-wide), with and without the
    desynchronised-batch hint (two instantiations of the 32-wide kernel, same results)."""
    _need_gpu()
    from finrl_amd import StockPanel
    from oracle.stock import StockOracle, lib, _p
    E, T = 200, 30
    close, tech, risk = _random_panel(5, T, N, K)
    rng = np.random.default_rng(9)
    kw = dict(hmax=100, initial_amount=300_000, turbulence_threshold=50.0)
    orc = StockOracle(close, tech, risk, n_envs=E, **kw)
    env = _make_env(StockPanel(close, tech, risk), E, auto_reset=True, **kw)
    env.hint_desynchronised(hint)
    orc.reset()
    env.reset()
    for s in range(80):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        o_obs, o_rew, o_done, _ = orc.vec_step(a)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done)
        if s in (3, 7, 12, 20):
            m = rng.random(E) < 0.3
            exp = g_obs.cpu().numpy().copy()
            for e in np.nonzero(m)[0]:
                row = np.empty(orc.D)
                lib().stock_oracle_reset_env(orc._h, C.c_int(int(e)), _p(row))
                exp[e] = row.astype(np.float32)
            g = env.reset(torch.from_numpy(m.astype(np.uint8)).cuda()).cpu().numpy()
            np.testing.assert_array_equal(g, exp)    # unmasked rows untouched
            st, os_ = env.state_numpy(), orc.state()
            np.testing.assert_array_equal(st["day"], os_["day"])
            np.testing.assert_array_equal(st["price_day"], os_["price_day"])
    st, os_ = env.state_numpy(), orc.state()
    assert len(np.unique(st["day"])) > 1
    np.testing.assert_array_equal(st["cash"], os_["cash"])
    np.testing.assert_array_equal(st["shares"], os_["shares"])


def test_sb3_adapter_protocol():
    """VecEnv-shaped adapter: shapes, dtypes, auto-reset and terminal_observation."""
    _need_gpu()
    fx = StockFixture("turbulence")
    E = 5
    env = _make_env(fx, E).as_sb3_vec_env()
    orc = fx.make_oracle(n_envs=E)
    obs = env.reset()
    o_obs = orc.reset()
    assert obs.dtype == np.float32 and obs.shape == (E, fx.D)
    np.testing.assert_array_equal(obs, o_obs.astype(np.float32))
    saw_done = False
    for s in range(fx.S):
        a = np.broadcast_to(fx.actions[s], (E, fx.N)).copy()
        obs, rew, done, infos = env.step(a)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        assert rew.dtype == np.float32 and done.dtype == bool and len(infos) == E
        np.testing.assert_array_equal(obs, o_obs.astype(np.float32))
        np.testing.assert_array_equal(rew, o_rew.astype(np.float32))
        np.testing.assert_array_equal(done, o_done)
        if done.any():
            saw_done = True
            for e in range(E):
                np.testing.assert_array_equal(infos[e]["terminal_observation"],
                                              o_term[e].astype(np.float32))
        else:
            assert all("terminal_observation" not in i for i in infos)
    assert saw_done


@pytest.mark.parametrize("name", ["tiefree", "ties", "turbulence", "cashbound", "n2", "prevstate",
                                  "n1", "n1_shares", "n1_turb", "n1_prevstate"])
def test_gym_facade_is_drop_in(name, capsys):
    """finrl_amd.meta.env_stock_trading.env_stocktrading.StockTradingEnv built from the SAME
    DataFrame and kwargs as the reference env: float64 list observations, rewards, asset /
    action memories and even the printed episode summary match the recorded reference run."""
    _need_gpu()
    import pandas as pd
    from finrl_amd.meta.env_stock_trading.env_stocktrading import StockTradingEnv
    fx = StockFixture(name)
    z = fx.z
    T, N, K = fx.T, fx.N, fx.K
    names = [f"ind{k}" for k in range(K)]
    cols = {"date": np.repeat([f"d{t:04d}" for t in range(T)], N),
            "tic": np.tile([f"TIC{i:03d}" for i in range(N)], T),
            "close": fx.close.reshape(-1)}
    for k, nme in enumerate(names):
        cols[nme] = fx.tech[:, k, :].reshape(-1)
    cols["turbulence"] = np.repeat(fx.risk, N)
    df = pd.DataFrame(cols)
    df.index = np.repeat(np.arange(T), N)
    kw = dict(df=df, stock_dim=N, hmax=fx.hmax, initial_amount=z["cfg_float"][5],
              num_stock_shares=fx.shares0.tolist(), buy_cost_pct=fx.buy_cost_pct,
              sell_cost_pct=fx.sell_cost_pct, reward_scaling=fx.reward_scaling,
              state_space=fx.D, action_space=N, tech_indicator_list=names,
              turbulence_threshold=fx.turbulence_threshold, print_verbosity=1, day=fx.day0)
    if not fx.initial:
        prev = [fx.cash0] + fx.close[0].tolist() + fx.shares0.tolist() + [0.0] * (K * N)
        kw.update(initial=False, previous_state=prev)
    env = StockTradingEnv(**kw)
    assert env.action_space.shape == (N,) and env.observation_space.shape == (fx.D,)
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == fx.D
    np.testing.assert_array_equal(np.asarray(obs, dtype=np.float64), resets[-1])
    tj = 0
    for s in range(fx.S):
        obs, rew, done, info = env.step(fx.actions[s].copy())
        np.testing.assert_array_equal(np.asarray(obs, dtype=np.float64), z["obs"][s])
        assert float(rew) == z["reward"][s] and bool(done) == bool(z["done"][s]) and info == {}
        assert env.trades == z["trades"][s] and env.cost == z["cost"][s]
        if not done:
            np.testing.assert_array_equal(env.actions_memory[-1], z["realised"][s])
        if done:
            np.testing.assert_array_equal(np.asarray(env.asset_memory), z[f"asset_memory_{tj}"])
            am = env.save_asset_memory()
            assert list(am.columns) == ["date", "account_value"] and len(am) == T
            acts = env.save_action_memory()
            # single ticker: a {"date", "actions"} frame (:536-542), else one column per ticker
            assert acts.shape == ((T - 1, N) if N > 1 else (T - 1, 2))
            tj += 1
            obs = env.reset()
            np.testing.assert_array_equal(np.asarray(obs, dtype=np.float64), resets[s])
    assert tj == 2
    printed = capsys.readouterr().out
    assert printed == str(z["printed"]), "episode summary text differs from the reference's"


@pytest.mark.parametrize("N,K", [(30, 8), (100, 2), (50, 3)])
def test_obs_row_pitch_changes_only_the_stride(N, K):
    """obs is a [E, D] view whose rows start on 64-byte boundaries by default
    (finenv_stock_set_obs_pitch); "packed" gives the reference's contiguous rows; `out=` takes any
    row stride.  Same values in all three, across terminal steps and auto-resets."""
    _need_gpu()
    from finrl_amd import StockPanel
    E, T = 200, 9
    close, tech, risk = _random_panel(3, T, N, K)
    panel = StockPanel(close, tech, risk)
    kw = dict(hmax=100, initial_amount=300_000, turbulence_threshold=50.0)
    a_env = _make_env(panel, E, **kw)
    p_env = _make_env(panel, E, obs_pitch="packed", **kw)
    o_env = _make_env(panel, E, obs_pitch=panel.D + 3, **kw)
    D = panel.D
    assert a_env.obs.stride(0) % 16 == 0 and a_env.obs.stride(0) >= D and a_env.obs.shape == (E, D)
    assert p_env.obs.is_contiguous() and o_env.obs.stride(0) == D + 3
    out_buf = torch.full((E, D + 40), -7.0, device="cuda")
    out = (out_buf[:, :D], torch.zeros(E, device="cuda"), torch.zeros(E, dtype=torch.uint8, device="cuda"))
    x_env = _make_env(panel, E, **kw)
    for env in (a_env, p_env, o_env, x_env):
        env.enable_terminal_obs()
    ref = p_env.reset().clone()
    assert torch.equal(a_env.reset(), ref) and torch.equal(o_env.reset(), ref)
    x_env.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    for s in range(2 * T + 3):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        po, pr, pdn, _ = p_env.step(a)
        for env in (a_env, o_env):
            o, r, d, _ = env.step(a)
            assert torch.equal(o, po) and torch.equal(r, pr) and torch.equal(d, pdn), s
            assert torch.equal(env.term_obs, p_env.term_obs)
        o, r, d, _ = x_env.step(a, out=out)
        assert torch.equal(o, po) and torch.equal(r, pr) and torch.equal(d, pdn)
        assert bool((out_buf[:, D:] == -7.0).all())          # nothing written past the row
        if s == 3:                                           # back to its own buffer and pitch
            o2 = x_env.observe()
            assert torch.equal(o2, po)
    assert bool((a_env._obs_buf[:, D:] == 0).all())
