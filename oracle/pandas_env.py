"""TEST INFRASTRUCTURE (oracle/): a reference-SHAPED single stock-trading env of this build's own
authorship -- pandas DataFrame backed, one `df.loc[day]` slice per step, Python-list state, Python
float money arithmetic -- used only as the `cpu_baseline_python` leg of bench.py (SURVEY.md 8d-ii:
"the apples-to-apples interpreter cost") and checked against the reference-generated fixtures in
tests/test_oracle_golden.py.  It restates the algorithm of
finrl/meta/env_stock_trading/env_stocktrading.py (step :220-357, reset :359-393, _sell_stock
:102-169, _buy_stock :171-213) with the canonical STABLE argsort; it is not
the product path and nothing under finrl_amd/ imports it.
"""
from __future__ import annotations

import numpy as np


def make_frame(close, tech, risk, risk_col="turbulence"):
    """[T,N] close, [T,K,N] tech, [T] risk -> the long frame the reference env is given: integer
    day index, N rows per day in ticker order (preprocessors.py:24-33 contract)."""
    import pandas as pd
    T, N = close.shape
    cols = {"tic": np.tile(np.arange(N), T), "close": np.asarray(close, np.float64).reshape(-1)}
    for k in range(tech.shape[1]):
        cols[f"tech{k}"] = np.asarray(tech[:, k, :], np.float64).reshape(-1)
    cols[risk_col] = np.repeat(np.asarray(risk, np.float64), N)
    df = pd.DataFrame(cols)
    df.index = np.repeat(np.arange(T), N)
    return df


class PandasStockEnv:
    def __init__(self, df, hmax=100, initial_amount=1_000_000, num_stock_shares=None,
                 buy_cost_pct=1e-3, sell_cost_pct=1e-3, reward_scaling=1e-4,
                 turbulence_threshold=None, risk_col="turbulence"):
        self.df = df
        self.tech_cols = [c for c in df.columns if c.startswith("tech")]
        self.n = int((df.index == df.index[0]).sum())
        self.n_days = int(df.index.max()) + 1
        self.hmax, self.cash0 = hmax, initial_amount
        self.shares0 = list(num_stock_shares) if num_stock_shares is not None else [0] * self.n
        self.c_buy, self.c_sell, self.scale = buy_cost_pct, sell_cost_pct, reward_scaling
        self.threshold, self.risk_col = turbulence_threshold, risk_col
        self.day = 0
        self.row = self.df.loc[self.day]
        self.cash, self.shares = initial_amount, list(self.shares0)
        self.turbulence, self.cost, self.trades, self.last_reward = 0, 0, 0, 0
        self.state = self._observe()

    def _observe(self):
        obs = [self.cash] + self.row["close"].values.tolist() + list(self.shares)
        for c in self.tech_cols:
            obs += self.row[c].values.tolist()
        return obs

    def reset(self):
        # the observation is rebuilt from the row held BEFORE the rewind (:361 vs :380-381)
        self.cash, self.shares = self.cash0, list(self.shares0)
        self.state = self._observe()
        self.day = 0
        self.row = self.df.loc[0]
        self.turbulence, self.cost, self.trades = 0, 0, 0
        return self.state

    def _asset(self, prices):
        return self.cash + sum(np.array(prices) * np.array(self.shares))

    def step(self, actions):
        n = self.n
        if self.day >= self.n_days - 1:                                     # :221
            return self.state, self.last_reward, True, {}
        act = (np.asarray(actions) * self.hmax).astype(int)                  # :304-305
        turbulent = self.threshold is not None and self.turbulence >= self.threshold
        if turbulent:
            act = np.array([-self.hmax] * n)
        prices = self.state[1:n + 1]                      # the prices the agent saw (stale after reset)
        flag = self.state[2 * n + 1:3 * n + 1] if self.tech_cols else [0.0] * n
        begin = self._asset(prices)
        order = np.argsort(act, kind="stable")
        for i in order[:int((act < 0).sum())]:                               # sells, most negative first
            p = prices[i]
            if turbulent:
                q = self.shares[i] if (p > 0 and self.shares[i] > 0) else None
            else:
                q = min(abs(int(act[i])), self.shares[i]) if (flag[i] != True and self.shares[i] > 0) \
                    else None                                                # noqa: E712 (fork quirk)
            if q is not None:
                self.cash += p * q * (1 - self.c_sell)
                self.shares[i] -= q
                self.cost += p * q * self.c_sell
                self.trades += 1
        for i in order[::-1][:int((act > 0).sum())]:                         # buys, largest first
            if turbulent or flag[i] == True:                                 # noqa: E712
                continue
            p = prices[i]
            q = min(self.cash // (p * (1 + self.c_buy)), int(act[i]))
            self.cash -= p * q * (1 + self.c_buy)
            self.shares[i] += q
            self.cost += p * q * self.c_buy
            self.trades += 1
        self.day += 1
        self.row = self.df.loc[self.day]
        if self.threshold is not None:
            self.turbulence = self.row[self.risk_col].values[0]
        self.state = self._observe()
        end = self._asset(self.state[1:n + 1])
        self.last_reward = (end - begin) * self.scale
        return self.state, self.last_reward, False, {}
