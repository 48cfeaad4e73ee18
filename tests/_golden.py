"""Helpers shared by the oracle and GPU parity tests: load a committed fixture
(tests/golden/stock_*.npz, produced by tests/golden/make_golden.py from the unmodified
reference) and build the matching oracle."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def stock_fixture_names():
    return sorted(os.path.basename(p)[len("stock_"):-4]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "stock_*.npz")))


class StockFixture:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, f"stock_{name}.npz"), allow_pickle=False)
        self.z = z
        ci = dict(zip(z["cfg_int_names"].tolist(), z["cfg_int"].tolist()))
        cf = dict(zip(z["cfg_float_names"].tolist(), z["cfg_float"].tolist()))
        self.T, self.N, self.K, self.S = ci["T"], ci["N"], ci["K"], ci["S"]
        self.hmax = ci["hmax"]
        self.use_turbulence = bool(ci["use_turbulence"])
        self.initial = bool(ci["initial"])
        self.day0 = ci["day0"]
        self.reset_first = bool(ci["reset_first"])
        self.cash0 = cf["cash0"]
        self.buy_cost_pct, self.sell_cost_pct = cf["buy_cost_pct"], cf["sell_cost_pct"]
        self.reward_scaling = cf["reward_scaling"]
        self.turbulence_threshold = cf["turbulence_threshold"] if self.use_turbulence else None
        self.D = 1 + 2 * self.N + self.K * self.N
        self.close, self.tech, self.risk = z["close"], z["tech"], z["risk"]
        self.actions = z["actions"]
        self.shares0 = z["shares0"]
        self.meta = z["meta"].tolist()

    def env_kwargs(self):
        """kwargs common to oracle.stock.StockOracle and the product's vector env."""
        return dict(hmax=self.hmax, initial_amount=self.cash0, num_stock_shares=self.shares0,
                    buy_cost_pct=self.buy_cost_pct, sell_cost_pct=self.sell_cost_pct,
                    reward_scaling=self.reward_scaling,
                    turbulence_threshold=self.turbulence_threshold, initial=self.initial,
                    day=self.day0)

    def make_oracle(self, n_envs=1, **over):
        from oracle.stock import StockOracle
        kw = self.env_kwargs()
        kw.update(over)
        return StockOracle(self.close, self.tech, self.risk, n_envs=n_envs, **kw)

    def sharpe(self, j):
        """sqrt(252)*mean/std(ddof=1) of pct_change(asset_memory) -- the quantity the
        reference computes at env_stocktrading.py:243-251 -- from the recorded list."""
        am = self.z[f"asset_memory_{j}"]
        r = am[1:] / am[:-1] - 1.0
        sd = r.std(ddof=1) if r.size > 1 else 0.0
        return np.nan if (r.size < 2 or sd == 0) else np.sqrt(252.0) * r.mean() / sd
