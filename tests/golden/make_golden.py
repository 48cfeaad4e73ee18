#!/usr/bin/env python3
"""Golden-vector generator (run ONLY in the build container, where /root/reference exists).

    python tests/golden/make_golden.py [name ...]

Runs the *unmodified* reference envs (imported through oracle/ref_harness.py) on small
seeded synthetic panels and scripted/random action sequences, and stores inputs and
outputs as compressed .npz fixtures next to this file.  The fixtures are data only
(inputs + the reference's outputs); no reference source is stored.

Oracle variants:
  O-raw    : reference as shipped.  Used where every step's scaled integer actions are
             distinct and non-zero, so np.argsort's tie order cannot matter.
  O-stable : same module with its np.argsort defaulting to kind="stable" (harness-side
             name substitution; reference files untouched).  Defines the canonical
             order for tie-heavy and turbulence sequences.
Each fixture records which variant produced it, the NumPy/pandas versions and the seed.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh  # noqa: E402


# ----------------------------------------------------------------------------- panels
def synth_panel(seed, T, N, K, *, flag_frac=0.0, fp32_prices=True, price0=100.0):
    """close [T,N], tech [T,K,N], risk [T] (all float64 values; fp32-representable
    when fp32_prices so a float32 device copy is bit-identical input)."""
    rng = np.random.default_rng(seed)
    close = price0 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    risk = np.abs(rng.normal(0, 30, T))
    if fp32_prices:
        close = close.astype(np.float32).astype(np.float64)
        tech = tech.astype(np.float32).astype(np.float64)
        risk = risk.astype(np.float32).astype(np.float64)
    if K:
        tech[:, 0, :][tech[:, 0, :] == 1.0] = 0.5
        if flag_frac > 0:  # plant the fork's "untradable" marker: first indicator == 1.0
            tech[:, 0, :][rng.random((T, N)) < flag_frac] = 1.0
    return close, tech, risk


def tiefree_actions(rng, S, N, hmax):
    """float32 actions whose (a*hmax).astype(int) are distinct and non-zero per step."""
    pool = np.concatenate([np.arange(-hmax, 0), np.arange(1, hmax + 1)])
    out = np.empty((S, N), dtype=np.float32)
    for s in range(S):
        k = rng.permutation(pool)[:N].astype(np.float64)
        out[s] = ((k + 0.5 * np.sign(k)) / hmax).astype(np.float32)
        got = (out[s] * hmax).astype(int)
        assert np.array_equal(got, k.astype(int)), (got, k)
    return out


# ----------------------------------------------------------------- primary stock env
def run_stock(name, *, seed, T, N, K, S, variant, actions="uniform", hmax=100,
              initial_amount=1_000_000, shares0=None, buy_cost_pct=1e-3, sell_cost_pct=1e-3,
              reward_scaling=1e-4, turbulence_threshold=None, flag_frac=0.0,
              previous_state=False, store_obs=True, zero_frac=0.0, fp32_prices=True,
              day0=0, reset_first=True):
    mod = rh.load_stocktrading()
    if variant == "O-stable":
        rh.stable_argsort_patch(mod)
    elif variant != "O-raw":
        raise ValueError(variant)
    rng = np.random.default_rng(seed + 1000)
    close, tech, risk = synth_panel(seed, T, N, K, flag_frac=flag_frac, fp32_prices=fp32_prices)
    names = [f"ind{k}" for k in range(K)]
    df = rh.make_stock_frame(close, tech, risk, names, risk_col="turbulence")
    if shares0 is None:
        shares0 = [0] * N
    shares0 = [int(x) for x in shares0]
    kwargs = dict(df=df, stock_dim=N, hmax=hmax, initial_amount=initial_amount,
                  num_stock_shares=list(shares0), buy_cost_pct=buy_cost_pct,
                  sell_cost_pct=sell_cost_pct, reward_scaling=reward_scaling,
                  state_space=1 + 2 * N + K * N, action_space=N, tech_indicator_list=names,
                  turbulence_threshold=turbulence_threshold, risk_indicator_col="turbulence",
                  print_verbosity=1, day=day0)
    prev = None
    if previous_state:
        prev_cash = float(initial_amount) * 0.73
        prev = [prev_cash] + close[0].tolist() + list(shares0) + [0.0] * (K * N)
        kwargs.update(initial=False, previous_state=prev)
    if actions == "uniform":
        act = rng.uniform(-1, 1, (S, N)).astype(np.float32)
        if zero_frac > 0:
            act[rng.random((S, N)) < zero_frac] = 0.0
    elif actions == "tiefree":
        act = tiefree_actions(rng, S, N, hmax)
    else:
        raise ValueError(actions)

    D = 1 + 2 * N + K * N
    rec = dict(obs=[], reward=[], done=[], cash=[], shares=[], cost=[], trades=[],
               turbulence=[], day=[], realised=[], reset_step=[], reset_obs=[],
               term_asset_memory=[], term_step=[])
    printed = io.StringIO()
    cwd = os.getcwd()
    os.makedirs("/tmp/golden_work/results", exist_ok=True)
    os.chdir("/tmp/golden_work")
    try:
        with contextlib.redirect_stdout(printed):
            env = mod.StockTradingEnv(**kwargs)
            obs0 = np.asarray(env.state, dtype=np.float64)
            if reset_first:   # what get_sb_env() does (env_stocktrading.py:549-552)
                rec["reset_step"].append(-1)
                rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float64))
            for s in range(S):
                a_in = act[s].copy()
                n_act_before = len(env.actions_memory)
                obs, rew, done, info = env.step(a_in)
                rec["obs"].append(np.asarray(obs, dtype=np.float64))
                rec["reward"].append(float(rew))
                rec["done"].append(bool(done))
                rec["cash"].append(float(env.state[0]))
                rec["shares"].append(np.asarray(env.state[1 + N:1 + 2 * N], dtype=np.float64))
                rec["cost"].append(float(env.cost))
                rec["trades"].append(int(env.trades))
                rec["turbulence"].append(float(env.turbulence))
                rec["day"].append(int(env.day))
                if len(env.actions_memory) > n_act_before:
                    rec["realised"].append(np.asarray(env.actions_memory[-1], dtype=np.int64))
                else:
                    rec["realised"].append(np.zeros(N, dtype=np.int64))
                if done:
                    rec["term_step"].append(s)
                    rec["term_asset_memory"].append(np.asarray(env.asset_memory, dtype=np.float64))
                    rec["reset_step"].append(s)
                    rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float64))
    finally:
        os.chdir(cwd)

    shares = np.stack(rec["shares"])
    assert np.array_equal(shares, np.round(shares)), "share counts must be integral"
    out = dict(
        # ---- inputs
        close=close, tech=tech, risk=risk, actions=act,
        cfg_int=np.array([T, N, K, S, hmax, int(turbulence_threshold is not None),
                          int(not previous_state), day0, int(reset_first)], dtype=np.int64),
        cfg_int_names=np.array(["T", "N", "K", "S", "hmax", "use_turbulence", "initial",
                                "day0", "reset_first"]),
        cfg_float=np.array([initial_amount if not previous_state else prev[0], buy_cost_pct,
                            sell_cost_pct, reward_scaling,
                            turbulence_threshold if turbulence_threshold is not None else 0.0,
                            initial_amount], dtype=np.float64),
        cfg_float_names=np.array(["cash0", "buy_cost_pct", "sell_cost_pct", "reward_scaling",
                                  "turbulence_threshold", "ctor_initial_amount"]),
        shares0=np.asarray(shares0, dtype=np.int64),
        # ---- reference outputs
        ctor_obs=obs0,
        reward=np.asarray(rec["reward"]), done=np.asarray(rec["done"]),
        cash=np.asarray(rec["cash"]), shares=shares.astype(np.int64),
        cost=np.asarray(rec["cost"]), trades=np.asarray(rec["trades"], dtype=np.int64),
        turbulence=np.asarray(rec["turbulence"]), day=np.asarray(rec["day"], dtype=np.int64),
        realised=np.stack(rec["realised"]),
        reset_step=np.asarray(rec["reset_step"], dtype=np.int64),
        reset_obs=(np.stack(rec["reset_obs"]) if rec["reset_obs"] else np.zeros((0, D))),
        term_step=np.asarray(rec["term_step"], dtype=np.int64),
        printed=np.array(printed.getvalue()),
        meta=np.array([f"variant={variant}", f"seed={seed}", f"numpy={np.__version__}",
                       f"pandas={__import__('pandas').__version__}",
                       "source=finrl/meta/env_stock_trading/env_stocktrading.py (unmodified)"]),
    )
    for j, am in enumerate(rec["term_asset_memory"]):
        out[f"asset_memory_{j}"] = am
    if store_obs:
        out["obs"] = np.stack(rec["obs"])
    path = os.path.join(HERE, f"stock_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)  steps={S} "
          f"episodes_done={len(rec['term_step'])} final_trades={rec['trades'][-1]}")
    return out


STOCK_SCENARIOS = {
    # tie-free, unmodified reference; 2.5 episodes => stale-reset quirk exercised twice
    "tiefree": dict(seed=11, T=24, N=30, K=8, S=60, variant="O-raw", actions="tiefree"),
    # the same tie-free run through O-stable must be identical (checked by the tests)
    "tiefree_stable": dict(seed=11, T=24, N=30, K=8, S=60, variant="O-stable", actions="tiefree"),
    # DOW30 x 8 shape, uniform actions (ties are the norm), cash binds after ~40 steps
    "ties": dict(seed=12, T=64, N=30, K=8, S=150, variant="O-stable", actions="uniform"),
    # cash-bound from step 0, non-zero starting shares, asymmetric costs, planted
    # untradable flags (first indicator == 1.0), different hmax / reward scaling
    "cashbound": dict(seed=13, T=30, N=30, K=8, S=75, variant="O-stable", actions="uniform",
                      hmax=50, initial_amount=50_000,
                      shares0=np.random.default_rng(5).integers(0, 20, 30),
                      buy_cost_pct=0.002, sell_cost_pct=0.0015, reward_scaling=1e-3,
                      flag_frac=0.1, zero_frac=0.1),
    # turbulence threshold: liquidation steps, buys suppressed, flag ignored when turbulent
    "turbulence": dict(seed=14, T=48, N=30, K=8, S=120, variant="O-stable", actions="uniform",
                       turbulence_threshold=40.0, flag_frac=0.05,
                       shares0=np.random.default_rng(6).integers(0, 10, 30)),
    # small ticker counts
    "n2": dict(seed=15, T=20, N=2, K=3, S=50, variant="O-stable", actions="uniform",
               initial_amount=20_000),
    "n3_tiefree": dict(seed=16, T=20, N=3, K=1, S=50, variant="O-raw", actions="tiefree",
                       initial_amount=30_000),
    # ensemble carry-over path: initial=False, previous_state (env_stocktrading.py:423-450)
    "prevstate": dict(seed=17, T=20, N=30, K=8, S=45, variant="O-stable", actions="uniform",
                      previous_state=True, shares0=np.random.default_rng(7).integers(0, 30, 30),
                      turbulence_threshold=60.0),
    # prices NOT rounded to fp32 (real-data like); engine keeps close in fp64
    "fp64prices": dict(seed=18, T=32, N=30, K=8, S=70, variant="O-stable", actions="uniform",
                       fp32_prices=False, initial_amount=200_000),
    # stepping straight after construction (no reset), from a non-zero start day
    "noreset_day3": dict(seed=19, T=16, N=5, K=2, S=30, variant="O-stable", actions="uniform",
                         day0=3, reset_first=False, initial_amount=10_000),
    # longer run, scalars only (no per-step obs): 2+ episodes of 400 days, NASDAQ-ish width
    # NASDAQ-100 width (128-wide kernel variant), turbulence, flags
    "n100": dict(seed=21, T=14, N=100, K=2, S=32, variant="O-stable", actions="uniform",
                 turbulence_threshold=45.0, flag_frac=0.03, initial_amount=400_000,
                 shares0=np.random.default_rng(8).integers(0, 8, 100)),
    # 50 tickers: the 64-wide kernel variant
    "n50": dict(seed=22, T=14, N=50, K=3, S=32, variant="O-stable", actions="uniform",
                turbulence_threshold=45.0, flag_frac=0.04, initial_amount=250_000,
                shares0=np.random.default_rng(9).integers(0, 8, 50)),
    "long": dict(seed=20, T=400, N=30, K=2, S=900, variant="O-stable", actions="uniform",
                 store_obs=False, turbulence_threshold=75.0),
    # single ticker: the reference's own single-stock branches (:417-422 `[0] * stock_dim` shares
    # whatever num_stock_shares says while asset_memory[0] still counts them :85-91 / :364-370;
    # :443-450, :470-476 scalar row access; :337-341 turbulence; :480-485 date)
    "n1": dict(seed=31, T=18, N=1, K=3, S=45, variant="O-stable", actions="uniform",
               initial_amount=5_000, hmax=30),
    "n1_shares": dict(seed=32, T=18, N=1, K=3, S=45, variant="O-stable", actions="uniform",
                      initial_amount=5_000, hmax=30, shares0=[7]),
    "n1_turb": dict(seed=33, T=18, N=1, K=2, S=45, variant="O-stable", actions="uniform",
                    initial_amount=4_000, hmax=25, shares0=[5], turbulence_threshold=30.0,
                    flag_frac=0.1),
    "n1_prevstate": dict(seed=34, T=18, N=1, K=3, S=45, variant="O-stable", actions="uniform",
                         initial_amount=6_000, hmax=30, shares0=[9], previous_state=True),
}


# ----------------------------------------------------------------- portfolio env
def run_portfolio(name, *, seed, T, N, K, S, initial_amount=1_000_000, lookback=8,
                  act_scale=1.0):
    """Unmodified reference StockPortfolioEnv (env_portfolio.py) on a synthetic frame with a
    per-day covariance object column (`cov_list`), as the reference's tutorials build it."""
    import pandas as pd
    mod = rh.load_portfolio()
    rng = np.random.default_rng(seed + 2000)
    close, tech, _ = synth_panel(seed, T, N, K, fp32_prices=False)
    rets = np.diff(np.log(close), axis=0, prepend=np.log(close[:1]))
    cov = np.empty((T, N, N))
    for t in range(T):
        w = rets[max(0, t - lookback):t + 1]
        cov[t] = np.cov(w.T) if len(w) > 1 else np.eye(N) * 1e-4
        if N == 1:
            cov[t] = np.atleast_2d(cov[t])
    names = [f"ind{k}" for k in range(K)]
    cols = {"date": np.repeat([f"d{t:04d}" for t in range(T)], N),
            "tic": np.tile([f"TIC{i:03d}" for i in range(N)], T),
            "close": close.reshape(-1)}
    for k, nme in enumerate(names):
        cols[nme] = tech[:, k, :].reshape(-1)
    df = pd.DataFrame(cols)
    df["cov_list"] = [cov[t] for t in range(T) for _ in range(N)]
    df.index = np.repeat(np.arange(T), N)
    act = (rng.uniform(0, 1, (S, N)) * act_scale).astype(np.float32)
    rec = dict(obs=[], reward=[], done=[], value=[], day=[], weights=[], reset_step=[],
               reset_obs=[])
    printed = io.StringIO()
    cwd = os.getcwd()
    os.makedirs("/tmp/golden_work/results", exist_ok=True)
    os.chdir("/tmp/golden_work")
    try:
        with contextlib.redirect_stdout(printed):
            env = mod.StockPortfolioEnv(df=df, stock_dim=N, hmax=100,
                                        initial_amount=initial_amount, transaction_cost_pct=0.001,
                                        reward_scaling=1e-4, state_space=N, action_space=N,
                                        tech_indicator_list=names)
            rec["reset_step"].append(-1)
            rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float64).reshape(-1))
            for s in range(S):
                n_w = len(env.actions_memory)
                obs, rew, done, info = env.step(act[s].copy())
                rec["obs"].append(np.asarray(obs, dtype=np.float64).reshape(-1))
                rec["reward"].append(float(rew))
                rec["done"].append(bool(done))
                rec["value"].append(float(env.portfolio_value))
                rec["day"].append(int(env.day))
                rec["weights"].append(np.asarray(env.actions_memory[-1], dtype=np.float32)
                                      if len(env.actions_memory) > n_w
                                      else np.zeros(N, np.float32))
                if done:
                    rec["reset_step"].append(s)
                    rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float64).reshape(-1))
    finally:
        os.chdir(cwd)
    out = dict(close=close, cov=cov, tech=tech, actions=act,
               cfg_int=np.array([T, N, K, S], dtype=np.int64),
               cfg_float=np.array([initial_amount], dtype=np.float64),
               obs=np.stack(rec["obs"]), reward=np.asarray(rec["reward"]),
               done=np.asarray(rec["done"]), value=np.asarray(rec["value"]),
               day=np.asarray(rec["day"], dtype=np.int64), weights=np.stack(rec["weights"]),
               reset_step=np.asarray(rec["reset_step"], dtype=np.int64),
               reset_obs=np.stack(rec["reset_obs"]),
               meta=np.array(["variant=O-raw", f"seed={seed}", f"numpy={np.__version__}",
                              "source=finrl/meta/env_portfolio_allocation/env_portfolio.py "
                              "(unmodified)"]))
    path = os.path.join(HERE, f"portfolio_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) steps={S} "
          f"dones={int(np.sum(rec['done']))} final_value={rec['value'][-1]:.2f}")
    return out


PORTFOLIO_SCENARIOS = {
    "dow30": dict(seed=31, T=20, N=30, K=8, S=48),
    "n5": dict(seed=32, T=16, N=5, K=2, S=40, initial_amount=50_000, act_scale=3.0),
    "n2k1": dict(seed=33, T=12, N=2, K=1, S=30, initial_amount=1_000),
}


# ----------------------------------------------------------------- multi-crypto env
def run_crypto(name, *, seed, T, N, W, S, lookback=1, initial_capital=1e6, price0=None,
               gamma=0.99, buy_cost_pct=1e-3, sell_cost_pct=1e-3, act_scale=1.0):
    """Unmodified reference CryptoEnv (env_multiple_crypto.py), float64 price/tech arrays."""
    mod = rh.load_multiple_crypto()
    rng = np.random.default_rng(seed + 3000)
    if price0 is None:
        price0 = 10.0 ** rng.uniform(-1, 4.5, N)
    price = np.asarray(price0) * np.exp(np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    act = (rng.uniform(-1, 1, (S, N)) * act_scale).astype(np.float32)
    act[rng.random((S, N)) < 0.1] = 0.0
    env = mod.CryptoEnv({"price_array": price, "tech_array": tech}, lookback=lookback,
                        initial_capital=initial_capital, buy_cost_pct=buy_cost_pct,
                        sell_cost_pct=sell_cost_pct, gamma=gamma)
    rec = dict(obs=[], reward=[], done=[], cash=[], stocks=[], total_asset=[], gamma_return=[],
               time=[], reset_step=[-1], reset_obs=[np.asarray(env.reset(), dtype=np.float32)])
    for s in range(S):
        obs, rew, done, info = env.step(act[s].copy())
        assert info is None
        rec["obs"].append(np.asarray(obs, dtype=np.float32))
        rec["reward"].append(float(rew))
        rec["done"].append(bool(done))
        rec["cash"].append(float(env.cash))
        rec["stocks"].append(np.asarray(env.stocks, dtype=np.float32).copy())
        rec["total_asset"].append(float(env.total_asset))
        rec["gamma_return"].append(float(env.gamma_return))
        rec["time"].append(int(env.time))
        if done:
            rec["reset_step"].append(s)
            rec["reset_obs"].append(np.asarray(env.reset(), dtype=np.float32))
    out = dict(price=price, tech=tech, actions=act, norm=np.asarray(env.action_norm_vector),
               cfg_int=np.array([T, N, W, S, lookback], dtype=np.int64),
               cfg_float=np.array([initial_capital, buy_cost_pct, sell_cost_pct, gamma]),
               obs=np.stack(rec["obs"]), reward=np.asarray(rec["reward"]),
               done=np.asarray(rec["done"]), cash=np.asarray(rec["cash"]),
               stocks=np.stack(rec["stocks"]), total_asset=np.asarray(rec["total_asset"]),
               gamma_return=np.asarray(rec["gamma_return"]),
               time=np.asarray(rec["time"], dtype=np.int64),
               reset_step=np.asarray(rec["reset_step"], dtype=np.int64),
               reset_obs=np.stack(rec["reset_obs"]),
               meta=np.array(["variant=O-raw", f"seed={seed}", f"numpy={np.__version__}",
                              "source=finrl/meta/env_cryptocurrency_trading/"
                              "env_multiple_crypto.py (unmodified)"]))
    path = os.path.join(HERE, f"crypto_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) steps={S} "
          f"dones={int(np.sum(rec['done']))} final_asset={rec['total_asset'][-1]:.2f}")
    return out


CRYPTO_SCENARIOS = {
    # 10 pairs x 4 indicators/pair (tutorials/3-Practical/FinRL_MultiCrypto_Trading.py:314,322)
    "pairs10": dict(seed=41, T=40, N=10, W=40, S=90),
    "lookback3": dict(seed=42, T=30, N=4, W=8, S=70, lookback=3, initial_capital=1e5,
                      price0=[30000.0, 2000.0, 0.5, 95.0]),
    "n1": dict(seed=43, T=20, N=1, W=3, S=45, initial_capital=5e4, price0=[123.4],
               buy_cost_pct=0.002, sell_cost_pct=0.0005, gamma=0.97),
    "n9_poor": dict(seed=44, T=25, N=9, W=5, S=60, initial_capital=2e3, act_scale=3.0),
}


# ----------------------------------------------------------------- array-state stock env (_np)
_TAG = {float: 0, np.float32: 1, np.float64: 2}


def _tag(x):
    return _TAG[type(x)] if type(x) in _TAG else {"float32": 1, "float64": 2}[np.asarray(x).dtype.name]


def run_stocknp(name, *, seed, T, N, K, S, if_train=False, initial_capital=1e6, max_stock=1e2,
                turb_scale=60.0, buy_cost_pct=1e-3, sell_cost_pct=1e-3, gamma=0.99, nas100=False,
                data_gap=1, turbulence_thresh=99):
    """Unmodified reference env_stocktrading_np.StockTradingEnv (NumPy-version dependent mixed
    float32/float64 arithmetic: the dtype of every scalar is recorded next to its value).
    nas100=True: env_nas100_wrds.StockEnvNAS100 instead -- the same step on float32 arrays handed
    over directly (cwd=None, if_eval=True: rows [0:211210:data_gap]), a random start state at
    every reset and an observation that shows max(amount, 1e4)."""
    rng = np.random.default_rng(seed + 4000)
    raw = {}
    if nas100:
        mod = rh._fresh_import("finrl.meta.env_stock_trading.env_nas100_wrds")
        Traw = T * data_gap
        raw["raw_price"] = (100 * np.exp(np.cumsum(rng.normal(0, 0.01, (Traw, N)), axis=0))).astype(np.float32)
        raw["raw_tech"] = rng.normal(0, 50, (Traw, N * K)).astype(np.float32)
        raw["raw_turb"] = np.abs(rng.normal(0, turb_scale, Traw)).astype(np.float32)
        env = mod.StockEnvNAS100(cwd=None, price_ary=raw["raw_price"], tech_ary=raw["raw_tech"],
                                 turbulence_ary=raw["raw_turb"], gamma=gamma,
                                 turbulence_thresh=turbulence_thresh, max_stock=max_stock,
                                 initial_capital=initial_capital, buy_cost_pct=buy_cost_pct,
                                 sell_cost_pct=sell_cost_pct, data_gap=data_gap, if_eval=True)
        env.stocks_cool_down = None                      # (this class calls it stocks_cd)
        price, tech, turb = (raw["raw_price"][::data_gap], raw["raw_tech"][::data_gap],
                             raw["raw_turb"][::data_gap])
        assert np.array_equal(env.price_ary, price) and env.max_step == T - 1
        if_train = True                                  # every reset draws its start state
    else:
        mod = rh.load_stocktrading_np()
        price = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
        tech = rng.normal(0, 50, (T, N * K))
        turb = np.abs(rng.normal(0, turb_scale, T))
        cfg = {"price_array": price, "tech_array": tech, "turbulence_array": turb,
               "if_train": if_train}
        env = mod.StockTradingEnv(cfg, initial_capital=initial_capital, max_stock=max_stock,
                                  buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct, gamma=gamma)
    act = rng.uniform(-1, 1, (S, N)).astype(np.float32)
    rec = {k: [] for k in ("obs", "reward", "reward_tag", "done", "amount", "amount_tag",
                           "total_asset", "ta_tag", "gamma_reward", "g_tag", "stocks",
                           "cool_down", "day", "episode_return")}
    resets = dict(step=[], obs=[], stocks0=[], amount0=[], amount0_tag=[])

    def do_reset(s):
        if if_train:
            mod.rd.seed(seed * 1000 + s + 7)
        o = env.reset()
        resets["step"].append(s)
        resets["obs"].append(np.asarray(o, np.float32))
        resets["stocks0"].append(np.asarray(env.stocks, np.float32).copy())
        resets["amount0"].append(float(env.amount))
        resets["amount0_tag"].append(_tag(env.amount))

    do_reset(-1)
    for s in range(S):
        obs, rew, done, info = env.step(act[s].copy())
        rec["obs"].append(np.asarray(obs, np.float32))
        rec["reward"].append(float(rew)); rec["reward_tag"].append(_tag(rew))
        rec["done"].append(bool(done))
        rec["amount"].append(float(env.amount)); rec["amount_tag"].append(_tag(env.amount))
        rec["total_asset"].append(float(env.total_asset)); rec["ta_tag"].append(_tag(env.total_asset))
        rec["gamma_reward"].append(float(env.gamma_reward)); rec["g_tag"].append(_tag(env.gamma_reward))
        rec["stocks"].append(np.asarray(env.stocks, np.float32).copy())
        rec["cool_down"].append(np.asarray(env.stocks_cd if nas100 else env.stocks_cool_down,
                                           np.float32).copy())
        rec["day"].append(int(env.day))
        rec["episode_return"].append(float(env.episode_return))
        if done:
            do_reset(s)
    out = dict(price_array=price, tech_array=tech, turbulence_array=turb, actions=act,
               cfg_int=np.array([T, N, K, S, int(if_train)], dtype=np.int64),
               cfg_float=np.array([initial_capital, max_stock, buy_cost_pct, sell_cost_pct, gamma]),
               obs=np.stack(rec["obs"]), done=np.asarray(rec["done"]),
               stocks=np.stack(rec["stocks"]), cool_down=np.stack(rec["cool_down"]),
               day=np.asarray(rec["day"], np.int64),
               reset_step=np.asarray(resets["step"], np.int64), reset_obs=np.stack(resets["obs"]),
               reset_stocks0=np.stack(resets["stocks0"]),
               reset_amount0=np.asarray(resets["amount0"]),
               reset_amount0_tag=np.asarray(resets["amount0_tag"], np.int64),
               meta=np.array(["variant=O-raw", f"seed={seed}", f"numpy={np.__version__}",
                              "dtype tags: 0=python float, 1=float32, 2=float64",
                              "source=finrl/meta/env_stock_trading/" +
                              ("env_nas100_wrds.py" if nas100 else "env_stocktrading_np.py") +
                              " (unmodified)"]))
    if nas100:
        out.update(raw, data_gap=np.array(data_gap, np.int64),
                   obs_amount_floor=np.array(1e4), turbulence_thresh=np.array(float(turbulence_thresh)),
                   reset_seed=np.asarray([seed * 1000 + s_ + 7 for s_ in resets["step"]], np.int64))
    for k in ("reward", "amount", "total_asset", "gamma_reward", "episode_return"):
        out[k] = np.asarray(rec[k], np.float64)
    for k in ("reward_tag", "amount_tag", "ta_tag", "g_tag"):
        out[k] = np.asarray(rec[k], np.int64)
    path = os.path.join(HERE, f"stocknp_{name}.npz")
    np.savez_compressed(path, **out)
    tags = sorted(set(rec["amount_tag"]))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) steps={S} "
          f"dones={int(np.sum(rec['done']))} amount dtypes seen={tags} "
          f"turbulent days={int((turb > 99).sum())}")
    return out


STOCKNP_SCENARIOS = {
    "eval_dow30": dict(seed=51, T=40, N=30, K=8, S=95),
    "eval_poor": dict(seed=52, T=30, N=30, K=8, S=70, initial_capital=3e4),
    "train_dow30": dict(seed=53, T=30, N=30, K=8, S=70, if_train=True),
    "eval_n3": dict(seed=54, T=25, N=3, K=2, S=60, initial_capital=5e3, max_stock=50.0,
                    buy_cost_pct=0.002, sell_cost_pct=0.0005, gamma=0.97, turb_scale=90.0),
    # StockEnvNAS100 (env_nas100_wrds.py): random start at every reset, obs shows max(amount, 1e4);
    # the poor one keeps the amount under that floor most of the time
    "nas100_dow30": dict(seed=55, T=30, N=30, K=8, S=70, nas100=True, data_gap=2, gamma=0.999,
                         turbulence_thresh=30, turb_scale=25.0),
    "nas100_poor": dict(seed=56, T=24, N=5, K=2, S=55, nas100=True, data_gap=4, gamma=0.999,
                        turbulence_thresh=30, turb_scale=25.0, initial_capital=3e4),
}


# ----------------------------------------------------------------- cash-penalty env
def run_cashpenalty(name, *, seed, T, N, S, cols=("open", "close", "high", "low", "volume"),
                    hmax=10, initial_amount=1e6, discrete_actions=False, shares_increment=1,
                    turbulence_threshold=None, patient=False, random_start=False,
                    cash_penalty_proportion=0.1, buy_cost_pct=3e-3, sell_cost_pct=3e-3,
                    act_scale=1.0):
    """Unmodified reference StockTradingEnvCashpenalty on a synthetic OHLCV frame."""
    import pandas as pd
    mod = _fresh_cashpenalty()
    rng = np.random.default_rng(seed + 5000)
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    data = {"open": close * rng.uniform(0.99, 1.01, (T, N)), "close": close,
            "high": close * 1.01, "low": close * 0.99,
            "volume": rng.integers(1e5, 1e6, (T, N)).astype(np.float64)}
    turb = np.abs(rng.normal(0, 30, T))
    tics = [f"TIC{i:03d}" for i in range(N)]
    frame = {"date": np.repeat([f"2020-{1 + t // 28:02d}-{1 + t % 28:02d}" for t in range(T)], N),
             "tic": np.tile(tics, T)}
    for c in set(cols) | {"close"}:
        frame[c] = data[c].reshape(-1)
    frame["turbulence"] = np.repeat(turb, N)
    df = pd.DataFrame(frame)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        env = mod.StockTradingEnvCashpenalty(
            df=df, buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct, hmax=hmax,
            discrete_actions=discrete_actions, shares_increment=shares_increment,
            turbulence_threshold=turbulence_threshold, print_verbosity=10 ** 9,
            initial_amount=initial_amount, daily_information_cols=list(cols),
            cache_indicator_data=True, cash_penalty_proportion=cash_penalty_proportion,
            random_start=random_start, patient=patient)
        act = (rng.uniform(-1, 1, (S, N)) * act_scale).astype(np.float32)
        rec = {k: [] for k in ("obs", "reward", "done", "coh", "holdings", "date_index",
                               "sum_trades")}
        resets = dict(step=[-1], obs=[np.asarray(env.reset(), np.float64)],
                      start=[env.starting_point])
        for s in range(S):
            obs, rew, done, info = env.step(act[s].copy())
            rec["obs"].append(np.asarray(obs, np.float64))
            rec["reward"].append(float(rew)); rec["done"].append(bool(done))
            rec["coh"].append(float(env.cash_on_hand))
            rec["holdings"].append(np.asarray(env.holdings, np.float64))
            rec["date_index"].append(int(env.date_index))
            rec["sum_trades"].append(float(env.sum_trades))
            if done:
                resets["step"].append(s)
                resets["obs"].append(np.asarray(env.reset(), np.float64))
                resets["start"].append(env.starting_point)
    info = np.stack([np.stack([data[c] for c in cols], axis=2)], axis=0)[0]     # [T, N, C]
    out = dict(close=close, info=info, turb=turb, actions=act,
               cfg_int=np.array([T, N, len(cols), S, int(discrete_actions), shares_increment,
                                 int(turbulence_threshold is not None), int(patient)], np.int64),
               cfg_float=np.array([hmax, buy_cost_pct, sell_cost_pct, initial_amount,
                                   cash_penalty_proportion,
                                   turbulence_threshold if turbulence_threshold is not None else 0.0]),
               obs=np.stack(rec["obs"]), reward=np.asarray(rec["reward"]),
               done=np.asarray(rec["done"]), coh=np.asarray(rec["coh"]),
               holdings=np.stack(rec["holdings"]),
               date_index=np.asarray(rec["date_index"], np.int64),
               sum_trades=np.asarray(rec["sum_trades"]),
               reset_step=np.asarray(resets["step"], np.int64), reset_obs=np.stack(resets["obs"]),
               reset_start=np.asarray(resets["start"], np.int64),
               meta=np.array(["variant=O-raw", f"seed={seed}", f"numpy={np.__version__}",
                              "source=finrl/meta/env_stock_trading/"
                              "env_stocktrading_cashpenalty.py (unmodified)"]))
    path = os.path.join(HERE, f"cashpenalty_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) steps={S} "
          f"dones={int(np.sum(rec['done']))} final_coh={rec['coh'][-1]:.2f}")
    return out


def _fresh_cashpenalty():
    rh.install()
    import importlib
    sys.modules.pop("finrl.meta.env_stock_trading.env_stocktrading_cashpenalty", None)
    return importlib.import_module("finrl.meta.env_stock_trading.env_stocktrading_cashpenalty")


CASHPENALTY_SCENARIOS = {
    "continuous": dict(seed=61, T=24, N=30, S=60, hmax=20_000, turbulence_threshold=55.0),
    "shortage": dict(seed=62, T=30, N=5, S=70, hmax=400_000, initial_amount=1e6),
    "patient": dict(seed=63, T=20, N=5, S=45, hmax=400_000, patient=True, act_scale=1.5),
    "discrete": dict(seed=64, T=20, N=8, S=45, hmax=30_000, discrete_actions=True,
                     shares_increment=5, cols=("close", "volume")),
    "randstart": dict(seed=65, T=30, N=3, S=60, hmax=50_000, random_start=True),
}


# ----------------------------------------------------------------- stop-loss env
def run_stoploss(name, *, seed, T, N, S, cols=("open", "close", "high", "low", "volume"),
                 hmax=10, initial_amount=1e6, discrete_actions=False, shares_increment=1,
                 stoploss_penalty=0.9, profit_loss_ratio=2, turbulence_threshold=None,
                 patient=False, random_start=False, cash_penalty_proportion=0.1,
                 buy_cost_pct=3e-3, sell_cost_pct=3e-3, act_scale=1.0, sigma=0.05):
    """Unmodified reference StockTradingEnvStopLoss on a synthetic (volatile) OHLCV frame."""
    import pandas as pd
    rh.install()
    import importlib
    sys.modules.pop("finrl.meta.env_stock_trading.env_stocktrading_stoploss", None)
    mod = importlib.import_module("finrl.meta.env_stock_trading.env_stocktrading_stoploss")
    rng = np.random.default_rng(seed + 6000)
    close = 50 * np.exp(np.cumsum(rng.normal(0, sigma, (T, N)), axis=0))
    data = {"open": close * rng.uniform(0.99, 1.01, (T, N)), "close": close,
            "high": close * 1.01, "low": close * 0.99,
            "volume": rng.integers(1e5, 1e6, (T, N)).astype(np.float64)}
    turb = np.abs(rng.normal(0, 30, T))
    tics = [f"TIC{i:03d}" for i in range(N)]
    frame = {"date": np.repeat([f"2020-{1 + t // 28:02d}-{1 + t % 28:02d}" for t in range(T)], N),
             "tic": np.tile(tics, T)}
    for c in set(cols) | {"close"}:
        frame[c] = data[c].reshape(-1)
    frame["turbulence"] = np.repeat(turb, N)
    df = pd.DataFrame(frame)
    buf = io.StringIO()
    vec_keys = ("holdings", "avg_buy_price", "n_buys", "closing_diff_avg_buy",
                "profit_sell_diff_avg_buy")
    with contextlib.redirect_stdout(buf):
        env = mod.StockTradingEnvStopLoss(
            df=df, buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct, hmax=hmax,
            discrete_actions=discrete_actions, shares_increment=shares_increment,
            stoploss_penalty=stoploss_penalty, profit_loss_ratio=profit_loss_ratio,
            turbulence_threshold=turbulence_threshold, print_verbosity=10 ** 9,
            initial_amount=initial_amount, daily_information_cols=list(cols),
            cache_indicator_data=True, cash_penalty_proportion=cash_penalty_proportion,
            random_start=random_start, patient=patient)
        act = (rng.uniform(-1, 1, (S, N)) * act_scale).astype(np.float32)
        rec = {k: [] for k in ("obs", "reward", "done", "coh", "date_index", "sum_trades",
                               "actual_num_trades") + vec_keys}
        resets = dict(step=[-1], obs=[np.asarray(env.reset(), np.float64)],
                      start=[env.starting_point])
        for s in range(S):
            obs, rew, done, info = env.step(act[s].copy())
            rec["obs"].append(np.asarray(obs, np.float64))
            rec["reward"].append(float(rew)); rec["done"].append(bool(done))
            rec["coh"].append(float(env.state_memory[-1][0]))
            rec["holdings"].append(np.asarray(env.state_memory[-1][1:N + 1], np.float64))
            for k in vec_keys[1:]:
                rec[k].append(np.asarray(getattr(env, k), np.float64).copy())
            rec["date_index"].append(int(env.date_index))
            rec["sum_trades"].append(float(env.sum_trades))
            rec["actual_num_trades"].append(float(env.actual_num_trades))
            if done:
                resets["step"].append(s)
                resets["obs"].append(np.asarray(env.reset(), np.float64))
                resets["start"].append(env.starting_point)
    reasons = sorted({r[2] for r in env.episode_history})
    info = np.stack([data[c] for c in cols], axis=2)                             # [T, N, C]
    out = dict(close=close, info=info, turb=turb, actions=act,
               cfg_int=np.array([T, N, len(cols), S, int(discrete_actions), shares_increment,
                                 int(turbulence_threshold is not None), int(patient)], np.int64),
               cfg_float=np.array([hmax, buy_cost_pct, sell_cost_pct, initial_amount,
                                   cash_penalty_proportion,
                                   turbulence_threshold if turbulence_threshold is not None else 0.0,
                                   stoploss_penalty, profit_loss_ratio]),
               obs=np.stack(rec["obs"]), reward=np.asarray(rec["reward"]),
               done=np.asarray(rec["done"]), coh=np.asarray(rec["coh"]),
               date_index=np.asarray(rec["date_index"], np.int64),
               sum_trades=np.asarray(rec["sum_trades"]),
               actual_num_trades=np.asarray(rec["actual_num_trades"]),
               reset_step=np.asarray(resets["step"], np.int64), reset_obs=np.stack(resets["obs"]),
               reset_start=np.asarray(resets["start"], np.int64),
               meta=np.array(["variant=O-raw", f"seed={seed}", f"numpy={np.__version__}",
                              "reasons=" + ",".join(reasons),
                              "source=finrl/meta/env_stock_trading/"
                              "env_stocktrading_stoploss.py (unmodified)"]))
    for k in vec_keys:
        out[k] = np.stack(rec[k])
    path = os.path.join(HERE, f"stoploss_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) steps={S} "
          f"dones={int(np.sum(rec['done']))} final_coh={rec['coh'][-1]:.2f} reasons={reasons}")
    return out


STOPLOSS_SCENARIOS = {
    "continuous": dict(seed=71, T=24, N=30, S=60, hmax=3_000, turbulence_threshold=55.0),
    "invested": dict(seed=72, T=30, N=5, S=70, hmax=150_000, sigma=0.06),
    "patient": dict(seed=73, T=20, N=5, S=45, hmax=400_000, patient=True, act_scale=1.5),
    "discrete": dict(seed=74, T=20, N=8, S=45, hmax=10_000, discrete_actions=True,
                     shares_increment=5, cols=("close", "volume"), turbulence_threshold=40.0),
    "randstart": dict(seed=75, T=30, N=3, S=60, hmax=30_000, random_start=True,
                      stoploss_penalty=0.95, profit_loss_ratio=3),
}


# ----------------------------------------------------------------- turbulence / covariance
def _load_preprocessors():
    """finrl.meta.preprocessor.preprocessors behind import shims for the absent third-party
    modules (stockstats, yfinance) -- the functions used here touch only numpy / pandas."""
    import importlib
    import types
    rh.install()
    sys.modules.setdefault("stockstats", types.SimpleNamespace(StockDataFrame=object))
    cfg = types.ModuleType("finrl.config")
    cfg.INDICATORS = []
    sys.modules.setdefault("finrl.config", cfg)
    yd = types.ModuleType("finrl.meta.preprocessor.yahoodownloader")
    yd.YahooDownloader = object
    sys.modules.setdefault("finrl.meta.preprocessor.yahoodownloader", yd)
    sys.modules["finrl"].config = cfg
    return importlib.import_module("finrl.meta.preprocessor.preprocessors")


def run_riskpre(name, *, seed, T, N, sigma=0.012, lookback=252, common=0.6):
    """FeatureEngineer.calculate_turbulence (unmodified) and the tutorial's cov_list lines
    (tutorials/2-Advance/FinRL_PortfolioAllocation_Explainable_DRL.py:160-172, the same
    pandas calls on the same frame) on a synthetic complete panel."""
    import pandas as pd
    pre = _load_preprocessors()
    rng = np.random.default_rng(seed + 7000)
    market = rng.normal(0, sigma, (T, 1))
    rets = common * market + rng.normal(0, sigma, (T, N)) * rng.uniform(0.5, 2.0, N)
    close = 80 * np.exp(np.cumsum(rets, axis=0))
    tics = [f"TIC{i:03d}" for i in range(N)]
    dates = pd.bdate_range("2015-01-01", periods=T).strftime("%Y-%m-%d")
    df = pd.DataFrame({"date": np.repeat(dates, N), "tic": np.tile(tics, T),
                       "close": close.reshape(-1)})
    fe = pre.FeatureEngineer(use_technical_indicator=False, use_turbulence=True)
    turb = fe.calculate_turbulence(df)
    assert list(turb["date"]) == list(dates)
    # tutorial :157-172
    d2 = df.sort_values(["date", "tic"], ignore_index=True)
    d2.index = d2.date.factorize()[0]
    cov_list = []
    for i in range(lookback, len(d2.index.unique())):
        data_lookback = d2.loc[i - lookback:i, :]
        price_lookback = data_lookback.pivot_table(index="date", columns="tic", values="close")
        return_lookback = price_lookback.pct_change().dropna()
        cov_list.append(return_lookback.cov().values)
    keep = np.unique(np.concatenate([np.arange(0, len(cov_list), 7), [len(cov_list) - 1]]))
    out = dict(close=close, turbulence=turb["turbulence"].to_numpy(np.float64),
               cov_index=keep.astype(np.int64), cov=np.stack([cov_list[k] for k in keep]),
               lookback=np.int64(lookback),
               meta=np.array([f"seed={seed}", f"numpy={np.__version__}",
                              f"pandas={pd.__version__}",
                              "source=finrl/meta/preprocessor/preprocessors.py:215-267 (unmodified)"
                              " + tutorial cov_list lines"]))
    path = os.path.join(HERE, f"riskpre_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB) T={T} N={N} "
          f"nonzero={int((out['turbulence'] > 0).sum())} max={out['turbulence'].max():.3f}")
    return out


RISKPRE_SCENARIOS = {
    "dow30": dict(seed=81, T=300, N=30),
    "small": dict(seed=82, T=280, N=6, sigma=0.02),
    "wide": dict(seed=83, T=262, N=100, common=0.8),
}


# ----------------------------------------------------------------- harness (caller loops)
def _df_columns(df):
    """DataFrame -> plain arrays (object columns holding arrays are stacked)."""
    out = {}
    for c in df.columns:
        v = df[c].tolist()
        if len(v) and isinstance(v[0], (np.ndarray, list)):
            out[c] = np.stack([np.asarray(x, np.float64) for x in v])
        elif len(v) and isinstance(v[0], str):
            out[c] = np.asarray(v)
        else:
            out[c] = np.asarray(v, np.float64)
    return out


def run_harness(name, *, kind, seed, T, N, K=8, **kw):
    """The reference's caller loops (tests/harness_loops.py restates them) run against the
    UNMODIFIED reference envs behind the DummyVecEnv stand-in of oracle/ref_harness.py:
      kind = sb3_stock        DRL_prediction over env_stocktrading.StockTradingEnv (O-stable)
      kind = sb3_cashpenalty  DRL_prediction over StockTradingEnvCashpenalty
      kind = sb3_stoploss     DRL_prediction over StockTradingEnvStopLoss
      kind = erl_stocknp      ElegantRL prediction loop over env_stocktrading_np.StockTradingEnv
    Stores the inputs and what the loops returned (DataFrame columns as arrays, printed text)."""
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness_loops as hl
    rng = np.random.default_rng(seed + 9000)
    base = rng.uniform(-1, 1, (T + 3, N)).astype(np.float32)
    printed = io.StringIO()
    out = dict(base=base, meta=np.array([f"kind={kind}", f"seed={seed}", f"numpy={np.__version__}",
                                         f"pandas={pd.__version__}"]))
    if kind == "sb3_stock":
        mod = rh.stable_argsort_patch(rh.load_stocktrading())
        thr = kw.get("turbulence_threshold")
        close, tech, risk = synth_panel(seed, T, N, K, flag_frac=kw.get("flag_frac", 0.0))
        names = [f"ind{k}" for k in range(K)]
        df = rh.make_stock_frame(close, tech, risk, names, risk_col="turbulence")
        ekw = dict(hmax=kw.get("hmax", 100), initial_amount=kw.get("initial_amount", 1_000_000),
                   buy_cost_pct=1e-3, sell_cost_pct=1e-3, reward_scaling=1e-4)
        with contextlib.redirect_stdout(printed):
            env = mod.StockTradingEnv(df=df, stock_dim=N, num_stock_shares=[0] * N,
                                      state_space=1 + 2 * N + K * N, action_space=N,
                                      tech_indicator_list=names, turbulence_threshold=thr,
                                      risk_indicator_col="turbulence", print_verbosity=1, **ekw)
            model = hl.ScriptedModel(base, 1 + np.arange(N))           # the close columns
            acct, acts = hl.drl_prediction(model, env)
        out.update(close=close, tech=tech, risk=risk,
                   cfg_int=np.array([T, N, K, ekw["hmax"], int(thr is not None)], np.int64),
                   cfg_float=np.array([ekw["initial_amount"], 1e-3, 1e-3, 1e-4,
                                       thr if thr is not None else 0.0]),
                   account_date=np.asarray(acct["date"].tolist()),
                   account_value=acct["account_value"].to_numpy(np.float64),
                   actions=acts.to_numpy(np.int64), action_date=np.asarray(acts.index.tolist()),
                   action_columns=np.asarray(acts.columns.tolist()),
                   action_index_name=np.array(str(acts.index.name)))
    elif kind in ("sb3_cashpenalty", "sb3_stoploss"):
        cols = list(kw.get("cols", ("open", "close", "high", "low", "volume")))
        sigma = kw.get("sigma", 0.01)
        close = 50 * np.exp(np.cumsum(rng.normal(0, sigma, (T, N)), axis=0))
        data = {"open": close * rng.uniform(0.99, 1.01, (T, N)), "close": close,
                "high": close * 1.01, "low": close * 0.99,
                "volume": rng.integers(1e5, 1e6, (T, N)).astype(np.float64)}
        turb = np.abs(rng.normal(0, 30, T))
        dates = [f"2020-{1 + t // 28:02d}-{1 + t % 28:02d}" for t in range(T)]
        frame = {"date": np.repeat(dates, N), "tic": np.tile([f"TIC{i:03d}" for i in range(N)], T)}
        for c in dict.fromkeys(cols + ["close"]):
            frame[c] = data[c].reshape(-1)
        frame["turbulence"] = np.repeat(turb, N)
        df = pd.DataFrame(frame)
        ekw = dict(buy_cost_pct=3e-3, sell_cost_pct=3e-3, hmax=kw.get("hmax", 20_000),
                   discrete_actions=kw.get("discrete_actions", False),
                   shares_increment=kw.get("shares_increment", 1),
                   turbulence_threshold=kw.get("turbulence_threshold"),
                   print_verbosity=kw.get("print_verbosity", 5),
                   initial_amount=kw.get("initial_amount", 1e6), daily_information_cols=cols,
                   cash_penalty_proportion=0.1, random_start=False,
                   patient=kw.get("patient", False))
        if kind == "sb3_cashpenalty":
            mod = _fresh_cashpenalty()
            cls = mod.StockTradingEnvCashpenalty
        else:
            rh.install()
            import importlib
            sys.modules.pop("finrl.meta.env_stock_trading.env_stocktrading_stoploss", None)
            mod = importlib.import_module("finrl.meta.env_stock_trading.env_stocktrading_stoploss")
            cls = mod.StockTradingEnvStopLoss
            ekw.update(stoploss_penalty=kw.get("stoploss_penalty", 0.9),
                       profit_loss_ratio=kw.get("profit_loss_ratio", 2))
        with contextlib.redirect_stdout(printed):
            env = cls(df=df, **ekw)
            # first information column of every asset
            model = hl.ScriptedModel(base, 1 + N + len(cols) * np.arange(N))
            try:
                acct, acts = hl.drl_prediction(model, env)
                raised = ""
            except IndexError as ex:      # episode cut short by a cash shortage: the reference's
                acct = acts = pd.DataFrame()    # loop then indexes an empty list (models.py:129)
                raised = f"IndexError: {ex}"
        out["raised"] = np.array(raised)
        out["model_steps"] = np.array(model.step, np.int64)
        info = np.stack([data[c] for c in cols], axis=2)
        out.update(close=close, info=info, turb=turb, cols=np.asarray(cols),
                   cfg_int=np.array([T, N, len(cols), int(ekw["discrete_actions"]),
                                     ekw["shares_increment"],
                                     int(ekw["turbulence_threshold"] is not None),
                                     int(ekw["patient"]), ekw["print_verbosity"]], np.int64),
                   cfg_float=np.array([ekw["hmax"], 3e-3, 3e-3, ekw["initial_amount"], 0.1,
                                       ekw["turbulence_threshold"] or 0.0,
                                       ekw.get("stoploss_penalty", 0.0),
                                       ekw.get("profit_loss_ratio", 0.0)]))
        for k, v in _df_columns(acct).items():
            out[f"account_{k}"] = v
        for k, v in _df_columns(acts).items():
            out[f"action_{k}"] = v
    elif kind == "stock_f64actions":
        # non-SB3 callers hand float64 arrays / Python lists: the reference scales them in the
        # CALLER's dtype (env_stocktrading.py:304-305), which truncates differently from float32
        mod = rh.stable_argsort_patch(rh.load_stocktrading())
        close, tech, risk = synth_panel(seed, T, N, K)
        names = [f"ind{k}" for k in range(K)]
        df = rh.make_stock_frame(close, tech, risk, names, risk_col="turbulence")
        act64 = np.round(rng.uniform(-1, 1, (T - 1, N)), 2)         # 0.29 * 100 -> 28.999999...
        with contextlib.redirect_stdout(printed):
            env = mod.StockTradingEnv(df=df, stock_dim=N, hmax=100, initial_amount=200_000,
                                      num_stock_shares=[0] * N, buy_cost_pct=1e-3,
                                      sell_cost_pct=1e-3, reward_scaling=1e-4,
                                      state_space=1 + 2 * N + K * N, action_space=N,
                                      tech_indicator_list=names, print_verbosity=10 ** 9)
            env.reset()
            rec = dict(reward=[], cash=[], shares=[], realised=[])
            for s_ in range(T - 1):
                a_in = act64[s_].copy() if s_ % 2 == 0 else act64[s_].tolist()   # array / list
                if isinstance(a_in, list):
                    a_in = np.array(a_in)      # (a list * int would repeat it: callers pass arrays)
                obs, rew, done, _ = env.step(a_in)
                rec["reward"].append(float(rew))
                rec["cash"].append(float(env.state[0]))
                rec["shares"].append(np.asarray(env.state[1 + N:1 + 2 * N], np.int64))
                rec["realised"].append(np.asarray(env.actions_memory[-1], np.int64))
        n_diff = int(((act64 * 100).astype(int) !=
                      (act64.astype(np.float32) * np.float32(100)).astype(int)).sum())
        assert n_diff > 0, "scenario must contain actions that truncate differently in float32"
        out.update(close=close, tech=tech, risk=risk, actions64=act64,
                   cfg_int=np.array([T, N, K, n_diff], np.int64),
                   reward=np.asarray(rec["reward"]), cash=np.asarray(rec["cash"]),
                   shares=np.stack(rec["shares"]), realised=np.stack(rec["realised"]))
    elif kind == "ensemble":
        # DRLEnsembleAgent's use of the env (models.py:213-230, :272-325): one validation window
        # (CSV dump read back for the Sharpe ratio), then two trade windows, the second seeded with
        # initial=False, previous_state=<render() of the first>.  Scalar costs (SURVEY.md headline 5).
        mod = rh.lenient_savefig_patch(rh.stable_argsort_patch(rh.load_stocktrading()))
        Tv, Tr = kw.get("Tv", 9), kw.get("Tr", 8)
        assert T == Tv + 2 * Tr
        thr = kw.get("turbulence_threshold")
        close, tech, risk = synth_panel(seed, T, N, K, flag_frac=kw.get("flag_frac", 0.0))
        names = [f"ind{k}" for k in range(K)]
        dates = [f"2021-{1 + t // 28:02d}-{1 + t % 28:02d}" for t in range(T)]

        def window(lo, hi):        # data_split: rows of [lo, hi), index re-factorized (preprocessors.py:24-33)
            return rh.make_stock_frame(close[lo:hi], tech[lo:hi], risk[lo:hi], names,
                                       risk_col="turbulence", dates=dates[lo:hi])
        ekw = dict(stock_dim=N, hmax=kw.get("hmax", 100), initial_amount=kw.get("initial_amount", 100_000),
                   num_stock_shares=[0] * N, buy_cost_pct=1e-3, sell_cost_pct=1e-3, reward_scaling=1e-4,
                   state_space=1 + 2 * N + K * N, action_space=N, tech_indicator_list=names,
                   print_verbosity=1)
        cwd = os.getcwd()
        work = "/tmp/golden_work_ens"
        import shutil
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(os.path.join(work, "results"))
        os.chdir(work)
        try:
            with contextlib.redirect_stdout(printed):
                val = window(0, Tv)
                model = hl.ScriptedModel(base, 1 + np.arange(N))
                val_env = rh._DummyVecEnv([lambda: mod.StockTradingEnv(
                    df=val, turbulence_threshold=thr, iteration=63, model_name="A2C",
                    mode="validation", **ekw)])
                val_obs = val_env.reset()
                hl.drl_validation(model, val, val_env, val_obs)
                sharpe = hl.get_validation_sharpe(63, "A2C")
                last1 = hl.ensemble_prediction(rh._DummyVecEnv, mod.StockTradingEnv, model,
                                               window(Tv, Tv + Tr), ekw, "ensemble", [], 126, thr, True)
                last2 = hl.ensemble_prediction(rh._DummyVecEnv, mod.StockTradingEnv, model,
                                               window(Tv + Tr, T), ekw, "ensemble", last1, 189, thr, False)
            files = {}
            for fn in sorted(os.listdir("results")):
                if fn.endswith(".csv"):
                    files[fn] = open(os.path.join("results", fn)).read()
        finally:
            os.chdir(cwd)
        out.update(close=close, tech=tech, risk=risk, dates=np.asarray(dates),
                   cfg_int=np.array([T, N, K, Tv, Tr, ekw["hmax"], int(thr is not None)], np.int64),
                   cfg_float=np.array([ekw["initial_amount"], 1e-3, 1e-3, 1e-4,
                                       thr if thr is not None else 0.0]),
                   sharpe=np.array(sharpe), last_state_1=np.asarray(last1, np.float64),
                   last_state_2=np.asarray(last2, np.float64),
                   csv_names=np.asarray(list(files)), csv_texts=np.asarray(list(files.values())),
                   model_steps=np.array(model.step, np.int64))
    elif kind == "erl_stocknp":
        mod = rh.load_stocktrading_np()
        price = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
        tech = rng.normal(0, 50, (T, N * K))
        turb = np.abs(rng.normal(0, 60, T))
        env = mod.StockTradingEnv({"price_array": price, "tech_array": tech,
                                   "turbulence_array": turb, "if_train": False})
        # the scaled price columns of the state (env_stocktrading_np.py:149-162)
        act = hl.scripted_act(base, 3 + np.arange(N))
        with contextlib.redirect_stdout(printed):
            assets, returns = hl.elegantrl_prediction(act, env)
        out.update(price_array=price, tech_array=tech, turbulence_array=turb,
                   cfg_int=np.array([T, N, K], np.int64),
                   episode_total_assets=np.asarray(assets, np.float64),
                   episode_returns=np.asarray(returns, np.float64))
    else:
        raise ValueError(kind)
    out["printed"] = np.array(printed.getvalue())
    path = os.path.join(HERE, f"harness_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)  kind={kind} "
          f"printed_lines={printed.getvalue().count(chr(10))}")
    return out


HARNESS_SCENARIOS = {
    "sb3_stock": dict(kind="sb3_stock", seed=81, T=26, N=30, K=8, initial_amount=150_000),
    "sb3_stock_turb": dict(kind="sb3_stock", seed=82, T=20, N=7, K=3, turbulence_threshold=35.0,
                           flag_frac=0.05, initial_amount=40_000),
    "sb3_cashpenalty": dict(kind="sb3_cashpenalty", seed=83, T=24, N=30, hmax=20_000,
                            turbulence_threshold=55.0),
    "sb3_cashpenalty_patient": dict(kind="sb3_cashpenalty", seed=84, T=20, N=5, hmax=400_000,
                                    patient=True, cols=("close", "volume")),
    "sb3_cashpenalty_shortage": dict(kind="sb3_cashpenalty", seed=85, T=20, N=5, hmax=600_000),
    "sb3_stoploss": dict(kind="sb3_stoploss", seed=86, T=30, N=8, hmax=60_000, sigma=0.05,
                         turbulence_threshold=70.0),
    "sb3_stoploss_patient": dict(kind="sb3_stoploss", seed=87, T=24, N=5, hmax=300_000,
                                 sigma=0.05, patient=True),
    "erl_stocknp": dict(kind="erl_stocknp", seed=88, T=40, N=30, K=8),
    "stock_f64actions": dict(kind="stock_f64actions", seed=89, T=16, N=30, K=2),
    "ensemble": dict(kind="ensemble", seed=90, T=25, N=6, K=3, Tv=9, Tr=8, initial_amount=30_000,
                     turbulence_threshold=45.0, flag_frac=0.03),
    "ensemble_dow30": dict(kind="ensemble", seed=91, T=30, N=30, K=8, Tv=10, Tr=10,
                           initial_amount=200_000),
}



def main(argv):
    names = argv or (list(STOCK_SCENARIOS) + ["portfolio:" + k for k in PORTFOLIO_SCENARIOS]
                     + ["crypto:" + k for k in CRYPTO_SCENARIOS]
                     + ["stocknp:" + k for k in STOCKNP_SCENARIOS]
                     + ["cashpenalty:" + k for k in CASHPENALTY_SCENARIOS]
                     + ["stoploss:" + k for k in STOPLOSS_SCENARIOS]
                     + ["riskpre:" + k for k in RISKPRE_SCENARIOS]
                     + ["harness:" + k for k in HARNESS_SCENARIOS])
    for n in names:
        if n in STOCK_SCENARIOS:
            run_stock(n, **STOCK_SCENARIOS[n])
        elif n.startswith("portfolio:") and n[10:] in PORTFOLIO_SCENARIOS:
            run_portfolio(n[10:], **PORTFOLIO_SCENARIOS[n[10:]])
        elif n.startswith("cashpenalty:") and n[12:] in CASHPENALTY_SCENARIOS:
            run_cashpenalty(n[12:], **CASHPENALTY_SCENARIOS[n[12:]])
        elif n.startswith("stoploss:") and n[9:] in STOPLOSS_SCENARIOS:
            run_stoploss(n[9:], **STOPLOSS_SCENARIOS[n[9:]])
        elif n.startswith("harness:") and n[8:] in HARNESS_SCENARIOS:
            run_harness(n[8:], **HARNESS_SCENARIOS[n[8:]])
        elif n.startswith("riskpre:") and n[8:] in RISKPRE_SCENARIOS:
            run_riskpre(n[8:], **RISKPRE_SCENARIOS[n[8:]])
        elif n.startswith("stocknp:") and n[8:] in STOCKNP_SCENARIOS:
            run_stocknp(n[8:], **STOCKNP_SCENARIOS[n[8:]])
        elif n.startswith("crypto:") and n[7:] in CRYPTO_SCENARIOS:
            run_crypto(n[7:], **CRYPTO_SCENARIOS[n[7:]])
        else:
            raise SystemExit(f"unknown scenario {n}")


if __name__ == "__main__":
    main(sys.argv[1:])
