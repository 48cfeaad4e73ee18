"""oracle/stoploss_oracle.c vs the committed outputs of the unmodified reference
StockTradingEnvStopLoss (tests/golden/stoploss_*.npz).

done / date_index / n_buys / market part of the observation: exact.  Money (cash, reward),
holdings and the average-buy-price family: rtol 1e-12 (BLAS ddot order, see the oracle header)."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("stoploss_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "stoploss_*.npz")))
VEC = ("holdings", "avg_buy_price", "n_buys", "closing_diff_avg_buy", "profit_sell_diff_avg_buy")


def oracle_kwargs(z):
    T, N, Cc, S, disc, inc, use_t, patient = z["cfg_int"].tolist()
    hmax, bc, sc, init, prop, thr, slp, plr = z["cfg_float"].tolist()
    return dict(buy_cost_pct=bc, sell_cost_pct=sc, hmax=hmax, discrete_actions=bool(disc),
                shares_increment=inc, turbulence_threshold=thr if use_t else None,
                initial_amount=init, cash_penalty_proportion=prop, patient=bool(patient),
                stoploss_penalty=slp, profit_loss_ratio=plr)


def test_fixtures_present():
    assert len(NAMES) >= 5


@pytest.mark.parametrize("name", NAMES)
def test_stoploss_oracle_matches_reference(name):
    from oracle.stoploss import StopLossOracle
    z = np.load(os.path.join(GOLDEN, f"stoploss_{name}.npz"), allow_pickle=False)
    T, N, Cc, S = z["cfg_int"].tolist()[:4]
    o = StopLossOracle(z["close"], z["info"], z["turb"], **oracle_kwargs(z))
    ri = 0
    obs = o.reset(z["reset_start"][ri])
    np.testing.assert_allclose(obs[0], z["reset_obs"][ri], rtol=1e-12)
    ri += 1
    nd = 0
    for s in range(S):
        obs, rew, done = o.step(z["actions"][s])
        st = o.state()
        assert done[0] == z["done"][s] and st["date_index"][0] == z["date_index"][s], s
        for k in VEC:
            np.testing.assert_allclose(st[k][0], z[k][s], rtol=1e-12, atol=1e-12,
                                       err_msg=f"{k} step {s}")
        np.testing.assert_array_equal(st["n_buys"][0], z["n_buys"][s])
        assert st["coh"][0] == pytest.approx(z["coh"][s], rel=1e-12), s
        assert rew[0] == pytest.approx(z["reward"][s], rel=1e-10, abs=1e-15), s
        np.testing.assert_array_equal(obs[0][1 + N:], z["obs"][s][1 + N:])
        np.testing.assert_allclose(obs[0][:1 + N], z["obs"][s][:1 + N], rtol=1e-12, atol=1e-12)
        assert st["sum_trades"][0] == pytest.approx(z["sum_trades"][s], rel=1e-6)
        if not done[0]:
            assert st["actual_num_trades"][0] == z["actual_num_trades"][s], s
        if done[0]:
            nd += 1
            obs = o.reset(z["reset_start"][ri])
            np.testing.assert_allclose(obs[0], z["reset_obs"][ri], rtol=1e-12)
            ri += 1
    assert nd >= 2


def test_zero_actions_keep_cash():
    """Upstream invariant (tests/environments/test_cash_penalty.py:29-52) on the sibling env."""
    from oracle.stoploss import StopLossOracle
    z = np.load(os.path.join(GOLDEN, "stoploss_continuous.npz"), allow_pickle=False)
    o = StopLossOracle(z["close"], z["info"], z["turb"], **oracle_kwargs(z))
    o.reset(0)
    N = z["close"].shape[1]
    for i in range(3):
        obs, rew, done = o.step(np.zeros((1, N), np.float32))
        assert obs[0][0] == z["cfg_float"][3] and obs[0][1:1 + N].sum() == 0 and not done[0]
        assert rew[0] == 0.0
