"""Full-size checks for the other BASELINE.json configs (per-GPU slices), in the style of
test_gpu_fullsize.py: size-independent properties over every env plus exact oracle parity on a
sampled subset (both ends of the batch, wave and block boundaries, random interior envs).

  configs[2]  65,536 StockPortfolioEnv, DOW30 x 8
  configs[3]  65,536 StockTradingEnv, 100 tickers x 8, turbulence threshold (1 of 8 GPUs' shard)
  configs[4]  32,768 CryptoEnv (262,144 / 8), 10 pairs x 4 indicators, PPO rollout buffers + GAE
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


def _sample(E, n=160, seed=0):
    s = np.random.default_rng(seed).choice(E, n, replace=False)
    s[:7] = [0, 63, 64, 127, 128, 255, E - 1]
    return np.unique(s)


def test_config2_portfolio_fullsize():
    _need_gpu()
    from finrl_amd.panel import PortfolioPanel
    from finrl_amd.vec_portfolio import VecStockPortfolioEnv
    from oracle.portfolio import PortfolioOracle
    from finrl_amd.riskpre import rolling_covariance
    E, T, N, K = 65_536, 120, 30, 8
    rng = np.random.default_rng(2)
    # the covariance state bench.py feeds: 252-day rolling window of the
    # returns, computed on the device (finenv_riskpre_rolling_cov), f32-rounded like the frame's
    # cov_list column; the series runs 252 days longer and the env sees the last T days
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T + 252, N)), axis=0))
    cov = rolling_covariance(close, lookback=252, device="cuda").cpu().numpy()
    close = close[252:]                      # cov_list starts at the first full window
    assert cov.shape == (T, N, N) and np.linalg.matrix_rank(cov[T // 2]) == N
    cov = cov.astype(np.float32).astype(np.float64)
    tech = rng.normal(0, 1, (T, K, N)).astype(np.float32).astype(np.float64)
    panel = PortfolioPanel(close, cov, tech)
    env = VecStockPortfolioEnv(panel, E, initial_amount=1e6, auto_reset=True)
    env.enable_weights()
    sample = _sample(E)
    orc = PortfolioOracle(close, cov, tech, n_envs=len(sample), initial_amount=1e6)
    obs = env.reset()
    np.testing.assert_array_equal(obs[sample].cpu().numpy(), orc.reset().astype(np.float32))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    gross = torch.from_numpy(close[1:] / close[:-1]).cuda()
    value = env.state["value"].clone()
    for s in range(50):
        a = torch.rand(E, N, generator=gen, device="cuda")
        day = env.state["day"].clone()
        obs, rew, done, _ = env.step(a)
        w = env.weights.double()
        # softmax weights: positive, sum to one (f32 rounding)
        assert float(w.min()) > 0 and float((w.sum(1) - 1).abs().max()) < 1e-5
        # value_{t+1} = value_t * (1 + sum((p_{t+1}/p_t - 1) * w))  (env_portfolio.py:183-188)
        growth = ((gross[day.long()] - 1) * w).sum(1)
        np.testing.assert_allclose(env.state["value"].cpu().numpy(),
                                   (value * (1 + growth)).cpu().numpy(), rtol=1e-9)
        value = env.state["value"].clone()
        assert torch.equal(env.state["day"], day + 1) and int(done.sum()) == 0
        o_obs, o_rew, o_done, _ = orc.vec_step(a[sample].cpu().numpy())
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_allclose(rew[sample].cpu().numpy(), o_rew, rtol=1e-5)   # f32 expf, 1 ulp
    np.testing.assert_allclose(env.state["value"][sample].cpu().numpy(), orc.state()["value"],
                               rtol=1e-5)


def test_config3_nasdaq100_shard_fullsize():
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    E, N = 65_536, 100
    close, tech, risk = bench.synth_panel(N=N)
    close, tech, risk = close[:400], tech[:400], risk[:400]
    kw = dict(bench.ENV_KW, turbulence_threshold=float(np.percentile(risk, 90)))
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **kw)
    sample = _sample(E, 128)
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **kw)
    obs = env.reset()
    np.testing.assert_array_equal(obs[sample].cpu().numpy(), orc.reset().astype(np.float32))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    close_t = torch.from_numpy(close).cuda()
    prev = env.total_asset().clone()
    saw_turbulent = False
    for s in range(40):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        turbulent = bool((env.state["turbulence"] >= kw["turbulence_threshold"]).all())
        obs, rew, done, _ = env.step(a)
        st = env.state
        assert int((st["holdings"] < 0).sum()) == 0 and float(st["cash"].min()) >= 0.0
        if turbulent:                                   # everything is sold, nothing bought
            saw_turbulent = True
            assert int(st["holdings"].sum()) == 0
        asset = st["cash"] + (close_t[st["day"].long()] * st["holdings"].T.double()).sum(1)
        np.testing.assert_allclose((rew.double() / kw["reward_scaling"]).cpu().numpy(),
                                   (asset - prev).cpu().numpy(), rtol=0, atol=4.0)
        prev = asset
        o_obs, o_rew, _, _ = orc.vec_step(a[sample].cpu().numpy())
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32),
                                      err_msg=f"step {s}")
        np.testing.assert_array_equal(rew[sample].cpu().numpy(), o_rew.astype(np.float32))
    assert saw_turbulent
    os_ = orc.state()
    np.testing.assert_array_equal(st["cash"][sample].cpu().numpy(), os_["cash"])
    np.testing.assert_array_equal(st["holdings"].T[sample].cpu().numpy(), os_["shares"])


def test_config4_crypto_shard_with_rollout_buffers():
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    from oracle.gae import gae as gae_reference
    E, T, N, W, n_steps = 32_768, 600, 10, 40, 32
    rng = np.random.default_rng(4)
    price = 10.0 ** rng.uniform(0, 4.5, N) * np.exp(np.cumsum(rng.normal(0, 5e-4, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E, auto_reset=True)
    sample = _sample(E, 128)
    orc = CryptoOracle(price, tech, n_envs=len(sample))
    obs0 = env.reset()
    np.testing.assert_array_equal(obs0[sample].cpu().numpy(), orc.reset())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    acts = [torch.rand(E, N, generator=gen, device="cuda") * 2 - 1 for _ in range(n_steps)]
    vals = [torch.rand(E, generator=gen, device="cuda") for _ in range(n_steps)]
    it = iter(range(n_steps))

    def policy(obs):
        t = next(it)
        return acts[t], vals[t], torch.zeros(E, device="cuda")

    buf = RolloutBuffer(n_steps, E, env.state_dim, N)
    last_obs = buf.collect(env, policy, obs0)
    # the kernels wrote straight into the [n_steps, E, .] tensors: compare slices with the oracle
    for t in range(n_steps):
        o_obs, o_rew, o_done, _ = orc.vec_step(acts[t][sample].cpu().numpy())
        np.testing.assert_array_equal(buf.obs[t + 1][sample].cpu().numpy(), o_obs, err_msg=f"t={t}")
        np.testing.assert_array_equal(buf.rewards[t][sample].cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(buf.dones[t][sample].cpu().numpy().astype(bool), o_done)
    assert torch.equal(last_obs, buf.obs[n_steps])
    # over every env: total_asset == cash + sum(stocks * price[time])  (:82).  (No sign property
    # holds: `cash // price` ignores the buy cost, so cash can dip below zero and the next buy
    # then "buys" a negative quantity -- the reference's behaviour, matched exactly above.)
    st = env.state
    price_t = torch.from_numpy(price).cuda()
    recomputed = st["cash"] + (st["stocks"].T.double() * price_t[st["time"].long()]).sum(1)
    np.testing.assert_allclose(st["total_asset"].cpu().numpy(), recomputed.cpu().numpy(), rtol=1e-12)
    last_v = torch.rand(E, generator=gen, device="cuda")
    adv, ret = buf.compute_returns_and_advantage(last_v, gamma=0.99, gae_lambda=0.95)
    ref_adv, ref_ret = gae_reference(buf.rewards[:, sample].cpu().numpy(),
                                     buf.values[:, sample].cpu().numpy(),
                                     buf.dones[:, sample].cpu().numpy(),
                                     last_v[sample].cpu().numpy(), 0.99, 0.95)
    np.testing.assert_allclose(adv[:, sample].cpu().numpy(), ref_adv, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret[:, sample].cpu().numpy(), ref_ret, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kind", ["cashpenalty", "stoploss"])
def test_cashpenalty_stoploss_fullsize_sampled_oracle(kind):
    """65,536 cash-penalty / stop-loss envs on per-env dates (pinned random starting points): the
    two-wave kernels with every block of the grid resident at once (LDS flag hand-off, speculated
    rows, auto-resets at different steps), exact against the oracle on a sampled subset."""
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv, VecStopLossEnv
    from oracle.cashpenalty import CashPenaltyOracle
    from oracle.stoploss import StopLossOracle
    E, T, N, Cc = 65_536, 48, 30, 5
    rng = np.random.default_rng(11)
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    info = rng.normal(0, 10, (T, N, Cc))
    turb = np.abs(rng.normal(0, 30, T))
    kw = dict(hmax=30_000, turbulence_threshold=70.0, patient=False, initial_amount=5e5,
              buy_cost_pct=0.002, sell_cost_pct=0.001, cash_penalty_proportion=0.15)
    cls, ocls = (VecCashPenaltyEnv, CashPenaltyOracle) if kind == "cashpenalty" else \
        (VecStopLossEnv, StopLossOracle)
    sample = _sample(E)
    env = cls(CashPenaltyPanel(close, info, turb), E, random_start=False, **kw)
    orc = ocls(close, info, turb, n_envs=len(sample), **kw)
    env.enable_terminal_obs()
    starts = rng.integers(0, T // 2, E).astype(np.int32)
    env.set_next_start(starts)
    np.testing.assert_array_equal(env.reset()[sample].cpu().numpy(),
                                  orc.reset(starts[sample]).astype(np.float32))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    n_done = 0
    for s in range(60):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        starts = rng.integers(0, T // 2, E).astype(np.int32)
        env.set_next_start(starts)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a[sample].cpu().numpy(), starts[sample])
        obs, rew, done, _ = env.step(a)
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), o_done, err_msg=f"{s}")
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(rew[sample].cpu().numpy(), o_rew.astype(np.float32))
        if o_done.any():
            np.testing.assert_array_equal(env.term_obs[sample].cpu().numpy()[o_done],
                                          o_term[o_done].astype(np.float32))
        n_done += int(done.sum())
        # every observation row is complete: cash column finite, no row left at the buffer's zeros
        assert bool(torch.isfinite(obs).all()) and float(obs[:, 0].min()) > 0
    st, os_ = env.state_numpy(), orc.state()
    for k in ("coh", "holdings", "date_index", "start", "episode"):
        np.testing.assert_array_equal(st[k][sample], os_[k], err_msg=k)
    assert n_done > E                                   # more than one episode end per env on average
