"""The single-env gym facades under ``finrl_amd.meta.*`` (same module paths, class names and
constructor keywords as the reference) replay the reference-recorded episodes: what a FinRL
user sees when switching the import."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_np_env_facade_replays_reference_episode():
    _need_gpu()
    from finrl_amd.meta.env_stock_trading.env_stocktrading_np import StockTradingEnv
    z = _load("stocknp_eval_n3.npz")
    T, N, K, S, if_train = z["cfg_int"].tolist()
    cap, ms, bc, sc, g = z["cfg_float"].tolist()
    env = StockTradingEnv({"price_array": z["price_array"], "tech_array": z["tech_array"],
                           "turbulence_array": z["turbulence_array"], "if_train": False},
                          gamma=g, max_stock=ms, initial_capital=cap, buy_cost_pct=bc,
                          sell_cost_pct=sc)
    assert (env.state_dim, env.action_dim, env.max_step) == (3 + 3 * N + N * K, N, T - 1)
    ri = 0
    np.testing.assert_array_equal(env.reset(), z["reset_obs"][ri])
    for s in range(S):
        obs, rew, done, info = env.step(z["actions"][s])
        np.testing.assert_array_equal(obs, z["obs"][s])
        assert rew == np.float32(z["reward"][s]) and done == bool(z["done"][s])
        assert env.day == z["day"][s] and env.amount == z["amount"][s]
        np.testing.assert_array_equal(env.stocks, z["stocks"][s])
        if done:
            assert env.episode_return == z["episode_return"][s]
            ri += 1
            np.testing.assert_array_equal(env.reset(), z["reset_obs"][ri])


def test_crypto_env_facade_replays_reference_episode():
    _need_gpu()
    from finrl_amd.meta.env_cryptocurrency_trading.env_multiple_crypto import CryptoEnv
    z = _load("crypto_btc_like.npz") if os.path.exists(os.path.join(GOLDEN, "crypto_btc_like.npz")) \
        else _load(sorted(f for f in os.listdir(GOLDEN) if f.startswith("crypto_"))[0])
    T, N, W, S, L = z["cfg_int"].tolist()
    cap, bc, sc, g = z["cfg_float"].tolist()
    env = CryptoEnv({"price_array": z["price"], "tech_array": z["tech"]}, lookback=L,
                    initial_capital=cap, buy_cost_pct=bc, sell_cost_pct=sc, gamma=g)
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    np.testing.assert_array_equal(env.reset(), resets[-1])
    for s in range(S):
        a = z["actions"][s].copy()
        obs, rew, done, info = env.step(a)
        assert info is None                                            # :90
        np.testing.assert_array_equal(a, (z["actions"][s] * z["norm"]).astype(np.float32))  # in place
        np.testing.assert_array_equal(obs, z["obs"][s])
        assert rew == np.float32(z["reward"][s]) and done == bool(z["done"][s])
        assert env.cash == z["cash"][s] and env.time == z["time"][s]
        if done:
            np.testing.assert_array_equal(env.reset(), resets[s])


def test_portfolio_env_facade_replays_reference_episode():
    _need_gpu()
    from finrl_amd.meta.env_portfolio_allocation.env_portfolio import StockPortfolioEnv
    from finrl_amd.panel import PortfolioPanel
    z = _load("portfolio_dow30.npz")
    T, N, K, S = z["cfg_int"].tolist()
    env = StockPortfolioEnv(PortfolioPanel(z["close"], z["cov"], z["tech"]), stock_dim=N, hmax=100,
                            initial_amount=float(z["cfg_float"][0]), transaction_cost_pct=0.001,
                            reward_scaling=1e-4, state_space=N, action_space=N,
                            tech_indicator_list=[f"t{k}" for k in range(K)])
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    np.testing.assert_array_equal(env.reset().reshape(-1), resets[-1].reshape(-1))
    for s in range(S):
        state, rew, done, info = env.step(z["actions"][s])
        assert done == bool(z["done"][s])
        np.testing.assert_array_equal(np.asarray(state).reshape(-1), z["obs"][s].reshape(-1))
        assert rew == pytest.approx(z["reward"][s], rel=1e-6)
        if done:
            assert len(env.save_asset_memory()) == len(env.asset_memory)
            assert env.save_action_memory().shape[1] == N
            np.testing.assert_array_equal(env.reset().reshape(-1), resets[s].reshape(-1))


@pytest.mark.parametrize("kind", ["cashpenalty", "stoploss"])
def test_dollar_env_facades_replay_reference_episode(kind):
    _need_gpu()
    import pandas as pd
    if kind == "cashpenalty":
        from finrl_amd.meta.env_stock_trading.env_stocktrading_cashpenalty import \
            StockTradingEnvCashpenalty as Env
    else:
        from finrl_amd.meta.env_stock_trading.env_stocktrading_stoploss import \
            StockTradingEnvStopLoss as Env
    z = _load(f"{kind}_discrete.npz")
    T, N, Cc, S, disc, inc, use_t, patient = z["cfg_int"].tolist()
    cf = z["cfg_float"].tolist()
    cols = ["close", "volume"]
    dates = [f"2020-{1 + t // 28:02d}-{1 + t % 28:02d}" for t in range(T)]
    frame = {"date": np.repeat(dates, N), "tic": np.tile([f"TIC{i:03d}" for i in range(N)], T),
             "turbulence": np.repeat(z["turb"], N)}
    for j, c in enumerate(cols):
        frame[c] = z["info"][:, :, j].reshape(-1)
    kw = dict(buy_cost_pct=cf[1], sell_cost_pct=cf[2], hmax=cf[0], discrete_actions=bool(disc),
              shares_increment=inc, turbulence_threshold=cf[5] if use_t else None,
              initial_amount=cf[3], daily_information_cols=cols, cash_penalty_proportion=cf[4],
              random_start=False, patient=bool(patient))
    if kind == "stoploss":
        kw.update(stoploss_penalty=cf[6], profit_loss_ratio=cf[7])
    env = Env(pd.DataFrame(frame), **kw)
    assert env.state_space == 1 + N + N * Cc
    np.testing.assert_allclose(env.reset(), z["reset_obs"][0], rtol=1e-12)
    nd = 0
    for s in range(S):
        state, rew, done, info = env.step(z["actions"][s])
        assert done == bool(z["done"][s]) and env.date_index == z["date_index"][s]
        np.testing.assert_allclose(state, z["obs"][s], rtol=1e-12, atol=1e-12)
        assert rew == pytest.approx(z["reward"][s], rel=1e-12, abs=1e-15)   # f64 (audit row)
        assert env.cash_on_hand == pytest.approx(z["coh"][s], rel=1e-12)
        np.testing.assert_allclose(env.holdings, z["holdings"][s], rtol=1e-12, atol=1e-12)
        if kind == "stoploss":
            np.testing.assert_allclose(env.avg_buy_price, z["avg_buy_price"][s], rtol=1e-12)
        if done:
            nd += 1
            env.reset()
    assert nd >= 2
    vec, obs0 = env.get_sb_env()
    assert obs0.shape == (1, env.state_space)
    o, r, d, infos = vec.step(np.zeros((1, N), np.float32))
    assert o.shape == (1, env.state_space) and r.shape == (1,) and isinstance(infos[0], dict)


def test_sb3_adapter_over_sibling_vec_envs():
    """The VecEnv-shaped adapter (numpy in / out, auto-reset, terminal_observation) over the crypto
    and cash-penalty batches: protocol shapes, and terminal observations at episode ends."""
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
    from finrl_amd.vec_crypto import VecCryptoEnv
    rng = np.random.default_rng(0)
    T, N, W, E = 12, 4, 6, 70
    price = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    crypto = VecCryptoEnv({"price_array": price, "tech_array": rng.normal(0, 100, (T, W))}, E)
    info = rng.normal(0, 1, (T, N, 2))
    cash = VecCashPenaltyEnv(CashPenaltyPanel(price, info), E, hmax=5_000, random_start=False)
    for env in (crypto, cash):
        venv = env.as_sb3_vec_env()
        obs = venv.reset()
        D = obs.shape[1]
        assert obs.shape == (E, D) and obs.dtype == np.float32 and venv.num_envs == E
        seen_done = 0
        for s in range(2 * T):
            obs, rew, done, infos = venv.step(rng.uniform(-1, 1, (E, N)).astype(np.float32))
            assert obs.shape == (E, D) and rew.shape == (E,) and done.dtype == bool
            assert len(infos) == E
            for i in np.nonzero(done)[0]:
                assert infos[i]["terminal_observation"].shape == (D,)
            seen_done += int(done.sum())
        assert seen_done >= E


@pytest.mark.parametrize("name", ["nas100_dow30", "nas100_poor"])
def test_nas100_facade_replays_reference_episode(name):
    """StockEnvNAS100 (env_nas100_wrds.py): constructor row selection (cwd=None, if_eval=True,
    data_gap), the random start state drawn from a seeded numpy.random at every reset, the
    max(amount, 1e4) observation, attributes the ElegantRL loops read."""
    _need_gpu()
    from finrl_amd.meta.env_stock_trading.env_nas100_wrds import StockEnvNAS100
    z = _load(f"stocknp_{name}.npz")
    T, N, K, S, _ = z["cfg_int"].tolist()
    cap, ms, bc, sc, g = z["cfg_float"].tolist()
    env = StockEnvNAS100(cwd=None, price_ary=z["raw_price"], tech_ary=z["raw_tech"],
                         turbulence_ary=z["raw_turb"], gamma=g,
                         turbulence_thresh=float(z["turbulence_thresh"]), max_stock=ms,
                         initial_capital=cap, buy_cost_pct=bc, sell_cost_pct=sc,
                         data_gap=int(z["data_gap"]), if_eval=True)
    assert (env.env_name, env.state_dim, env.action_dim, env.max_step, env.target_return) == \
        ("StockEnvNAS", 3 + 3 * N + N * K, N, T - 1, 2.2)
    np.testing.assert_array_equal(env.price_ary, z["price_array"])
    ri = 0

    def do_reset():
        nonlocal ri
        np.random.seed(int(z["reset_seed"][ri]))          # what the fixture generator seeded
        obs = env.reset()
        np.testing.assert_array_equal(obs, z["reset_obs"][ri])
        np.testing.assert_array_equal(env.stocks, z["reset_stocks0"][ri])
        assert env.amount == z["reset_amount0"][ri] and env.day == 0
        ri += 1

    do_reset()
    for s in range(S):
        obs, rew, done, info = env.step(z["actions"][s])
        np.testing.assert_array_equal(obs, z["obs"][s])
        assert rew == np.float32(z["reward"][s]) and done == bool(z["done"][s])
        assert env.day == z["day"][s] and env.amount == z["amount"][s]
        np.testing.assert_array_equal(env.stocks, z["stocks"][s])
        np.testing.assert_array_equal(env.stocks_cd, z["cool_down"][s])
        if done:
            assert env.episode_return == z["episode_return"][s]
            do_reset()
    assert ri == 3
