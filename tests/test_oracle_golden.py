"""CPU restatement (oracle/stock_oracle.c) vs the committed reference outputs.

Bar: holdings/day/done/trades exact; fp64 money quantities bit-exact
where the order of operations is fully pinned (cash, reward, cost, obs) -- we assert
exact equality, not a tolerance -- and <= 1e-9 rel for the running Sharpe.
"""
import re

import numpy as np
import pytest

from _golden import StockFixture, stock_fixture_names

NAMES = stock_fixture_names()


def _replay(fx, o):
    """Replay fixture actions through oracle `o` (E=1), yielding per-step tuples."""
    z = fx.z
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    if -1 in resets:
        obs = o.reset()
        np.testing.assert_array_equal(obs[0], resets[-1])
    term_j = 0
    for s in range(fx.S):
        obs, rew, done, real = o.step(fx.actions[s], want_realised=True)
        st = o.state()
        yield s, obs[0], rew[0], done[0], real[0], st
        if done[0]:
            stats = o.episode_stats()[0]
            am = z[f"asset_memory_{term_j}"]
            assert stats[0] == am[0]
            assert stats[1] == z["cash"][s] + sum(
                z["obs"][s][1:1 + fx.N] * z["shares"][s]) if "obs" in z.files else True
            assert stats[3] == z["cost"][s] and stats[4] == z["trades"][s]
            sh = fx.sharpe(term_j)
            if np.isnan(sh):
                assert np.isnan(stats[5])
            else:
                assert stats[5] == pytest.approx(sh, rel=1e-9, abs=1e-12)
            term_j += 1
            obs = o.reset()
            np.testing.assert_array_equal(obs[0], resets[s])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    fx = StockFixture(name)
    z = fx.z
    o = fx.make_oracle()
    assert o.D == fx.D
    n = 0
    for s, obs, rew, done, real, st in _replay(fx, o):
        assert done == z["done"][s], s
        assert st["day"][0] == z["day"][s], s
        np.testing.assert_array_equal(st["shares"][0], z["shares"][s], err_msg=f"step {s}")
        assert st["trades"][0] == z["trades"][s], s
        assert st["cash"][0] == z["cash"][s], (s, st["cash"][0], z["cash"][s])
        assert rew == z["reward"][s], (s, rew, z["reward"][s])
        assert st["cost"][0] == z["cost"][s], s
        assert st["turbulence"][0] == z["turbulence"][s], s
        np.testing.assert_array_equal(real, z["realised"][s], err_msg=f"step {s}")
        if "obs" in z.files:
            np.testing.assert_array_equal(obs, z["obs"][s], err_msg=f"step {s}")
        n += 1
    assert n == fx.S and z["done"].sum() >= 2


def test_tiefree_raw_equals_stable():
    """On tie-free sequences the unmodified reference (O-raw) and the stable-argsort
    variant (O-stable) must agree bit for bit -- this is what licenses using O-stable
    as the canonical order elsewhere."""
    a, b = StockFixture("tiefree").z, StockFixture("tiefree_stable").z
    for k in ("obs", "reward", "cash", "shares", "cost", "trades", "realised", "reset_obs"):
        np.testing.assert_array_equal(a[k], b[k])
    assert "variant=O-raw" in a["meta"].tolist() and "variant=O-stable" in b["meta"].tolist()


def test_printed_episode_summary():
    """The reference's own printed terminal summary (env_stocktrading.py:255-264) agrees
    with the oracle's episode_stats at print precision."""
    fx = StockFixture("ties")
    o = fx.make_oracle()
    blocks = str(fx.z["printed"]).split("=================================")
    o.reset()
    j = 0
    for s in range(fx.S):
        _, _, done = o.step(fx.actions[s])
        if done[0]:
            st = o.episode_stats()[0]
            txt = blocks[j]
            f = lambda key: float(re.search(key + r":\s*(-?[\d.]+)", txt).group(1))
            assert f("begin_total_asset") == pytest.approx(st[0], abs=6e-3)
            assert f("end_total_asset") == pytest.approx(st[1], abs=6e-3)
            assert f("total_reward") == pytest.approx(st[2], abs=6e-3)
            assert f("total_cost") == pytest.approx(st[3], abs=6e-3)
            assert int(f("total_trades")) == int(st[4])
            assert f("Sharpe") == pytest.approx(st[5], abs=6e-4)
            j += 1
            o.reset()
    assert j == 2


def test_vec_step_autoreset_matches_manual():
    fx = StockFixture("turbulence")
    a, b = fx.make_oracle(), fx.make_oracle()
    a.reset(); b.reset()
    for s in range(fx.S):
        obs, rew, done, term = a.vec_step(fx.actions[s])
        obs2, rew2, done2 = b.step(fx.actions[s])
        assert rew[0] == rew2[0] and done[0] == done2[0]
        if done2[0]:
            np.testing.assert_array_equal(term[0], obs2[0])
            obs2 = b.reset()
        np.testing.assert_array_equal(obs, obs2)


def test_zero_action_changes_nothing():
    """Invariant restated from the reference's tests/environments/test_cash_penalty.py:29-52."""
    fx = StockFixture("ties")
    o = fx.make_oracle(n_envs=3)
    o.reset()
    for _ in range(5):
        obs, rew, done = o.step(np.zeros((3, fx.N), np.float32))
        st = o.state()
        assert (st["cash"] == fx.cash0).all() and (st["shares"] == 0).all()
        assert (st["trades"] == 0).all() and (rew == 0).all()


def test_unaffordable_buy_zero_fill():
    """Invariant restated from test_cash_penalty.py:55-75: a buy that cash cannot cover
    fills zero shares (and, in this env, still counts as a trade, :197)."""
    fx = StockFixture("ties")
    o = fx.make_oracle(initial_amount=50.0)   # < one share of anything (~100)
    o.reset()
    obs, rew, done = o.step(np.ones((1, fx.N), np.float32))
    st = o.state()
    assert (st["shares"] == 0).all() and st["cash"][0] == 50.0 and st["trades"][0] == fx.N


@pytest.mark.parametrize("name", ["ties", "turbulence", "cashbound", "tiefree"])
def test_pandas_shaped_env_matches_reference(name):
    """oracle/pandas_env.py (the reference-SHAPED Python env timed as `cpu_baseline_python`) replays
    the reference-recorded episodes exactly: float64 observations, rewards, cash, shares, cost,
    trades, including the stale-row reset quirk."""
    from oracle.pandas_env import PandasStockEnv, make_frame
    fx = StockFixture(name)
    z = fx.z
    if not (fx.initial and fx.day0 == 0 and fx.reset_first):
        pytest.skip("plain constructor path only")
    env = PandasStockEnv(make_frame(fx.close, fx.tech, fx.risk), hmax=fx.hmax,
                         initial_amount=fx.cash0, num_stock_shares=fx.shares0.tolist(),
                         buy_cost_pct=fx.buy_cost_pct, sell_cost_pct=fx.sell_cost_pct,
                         reward_scaling=fx.reward_scaling,
                         turbulence_threshold=fx.turbulence_threshold)
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    np.testing.assert_array_equal(np.asarray(env.reset(), np.float64), resets[-1])
    for s in range(fx.S):
        obs, rew, done, _ = env.step(fx.actions[s])
        assert done == z["done"][s] and rew == z["reward"][s], s
        assert env.cash == z["cash"][s] and env.cost == z["cost"][s] and env.trades == z["trades"][s]
        np.testing.assert_array_equal(np.asarray(env.shares, np.float64), z["shares"][s])
        np.testing.assert_array_equal(np.asarray(obs, np.float64), z["obs"][s])
        if done:
            np.testing.assert_array_equal(np.asarray(env.reset(), np.float64), resets[s])
