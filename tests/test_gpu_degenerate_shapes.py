"""Smallest shapes the kernels accept: one env, one asset, two days, no indicator columns -- a single
partially filled wave per launch, panels of two rows, episodes that end on the first step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.mark.parametrize("E,T,N,K", [(1, 2, 1, 0), (1, 3, 2, 1), (3, 2, 30, 8), (2, 2, 100, 1)])
def test_stock_env_smallest_shapes(E, T, N, K):
    _need_gpu()
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    rng = np.random.default_rng(E + T + N)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    risk = np.abs(rng.normal(0, 30, T))
    kw = dict(hmax=50, initial_amount=20_000, turbulence_threshold=60.0)
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **kw)
    orc = StockOracle(close, tech, risk, n_envs=E, **kw)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset().astype(np.float32))
    for s in range(3 * T):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_done, _ = orc.vec_step(a)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs.astype(np.float32), err_msg=f"{s}")
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(done.cpu().numpy().astype(bool), o_done)


@pytest.mark.parametrize("kind", ["cashpenalty", "stoploss"])
@pytest.mark.parametrize("E,T,N,C", [(1, 2, 1, 0), (1, 3, 1, 1), (2, 4, 2, 3)])
def test_cashpenalty_stoploss_smallest_shapes(kind, E, T, N, C):
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv, VecStopLossEnv
    from oracle.cashpenalty import CashPenaltyOracle
    from oracle.stoploss import StopLossOracle
    rng = np.random.default_rng(E + T + N + C)
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    info = rng.normal(0, 10, (T, N, C))
    kw = dict(hmax=5_000, initial_amount=1e5)
    cls, ocls = (VecCashPenaltyEnv, CashPenaltyOracle) if kind == "cashpenalty" else \
        (VecStopLossEnv, StopLossOracle)
    env = cls(CashPenaltyPanel(close, info), E, random_start=False, **kw)
    orc = ocls(close, info, None, n_envs=E, **kw)
    starts = np.zeros(E, np.int32)
    env.set_next_start(starts)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset(starts).astype(np.float32))
    for s in range(3 * T):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        env.set_next_start(starts)
        o_obs, o_rew, o_done, _ = orc.vec_step(a, starts)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(done.cpu().numpy().astype(bool), o_done, err_msg=f"{s}")
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew.astype(np.float32))


def test_crypto_and_stocknp_single_env():
    _need_gpu()
    from finrl_amd.vec_crypto import VecCryptoEnv
    from finrl_amd.vec_stocknp import VecStockTradingEnvNP
    from oracle.crypto import CryptoOracle
    rng = np.random.default_rng(3)
    T, N, W = 8, 1, 2
    price = 100 * np.exp(np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, 1, lookback=1)
    orc = CryptoOracle(price, tech, n_envs=1, lookback=1)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset())
    for s in range(2 * T):
        a = rng.uniform(-1, 1, (1, N)).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_done, _ = orc.vec_step(a)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs, err_msg=f"{s}")
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(done.cpu().numpy().astype(bool), o_done)
    cfg = {"price_array": price.astype(np.float32), "tech_array": rng.normal(0, 50, (T, N * 2)).astype(np.float32),
           "turbulence_array": np.abs(rng.normal(0, 20, T)).astype(np.float32), "if_train": False}
    np_env = VecStockTradingEnvNP(cfg, 1)
    o = np_env.reset()
    assert o.shape == (1, 3 + 3 * N + 2 * N) and bool(torch.isfinite(o).all())
    for s in range(T + 2):
        o, r, d, _ = np_env.step(torch.rand(1, N, device="cuda") * 2 - 1)
        assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all())
