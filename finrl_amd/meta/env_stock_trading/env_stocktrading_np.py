"""Drop-in for ``finrl.meta.env_stock_trading.env_stocktrading_np.StockTradingEnv``
(env_stocktrading_np.py:9-169 in the reference tree): same constructor keywords, attributes
(``env_name, state_dim, action_dim, max_step, if_discrete, target_return, amount, stocks, day,
initial_total_asset, episode_return``) and ``reset() / step()`` protocol, one HIP launch per
step through the C ABI (finenv_stocknp_*).  Single-env facade over
:class:`finrl_amd.vec_stocknp.VecStockTradingEnvNP`; use ``make_vec`` for throughput."""
from __future__ import annotations

import numpy as np

from ...spaces import Box
from ...vec_stocknp import VecStockTradingEnvNP
from .._single import to_action_tensor


class StockTradingEnv:
    def __init__(self, config, initial_account=1e6, gamma=0.99, turbulence_thresh=99,
                 min_stock_rate=0.1, max_stock=1e2, initial_capital=1e6, buy_cost_pct=1e-3,
                 sell_cost_pct=1e-3, reward_scaling=2 ** -11, initial_stocks=None, device="cuda",
                 seed=0):
        self._kw = dict(gamma=gamma, turbulence_thresh=turbulence_thresh,
                        min_stock_rate=min_stock_rate, max_stock=max_stock,
                        initial_capital=initial_capital, buy_cost_pct=buy_cost_pct,
                        sell_cost_pct=sell_cost_pct, reward_scaling=reward_scaling,
                        initial_stocks=initial_stocks)
        self._config = config
        self._vec = VecStockTradingEnvNP(config, 1, auto_reset=False, device=device, seed=seed,
                                         **self._kw)
        v = self._vec
        self.price_ary, self.tech_ary = v.price_ary, v.tech_ary
        self.turbulence_ary, self.turbulence_bool = v.turbulence_ary, v.turbulence_bool
        self.gamma, self.max_stock, self.min_stock_rate = gamma, max_stock, min_stock_rate
        self.buy_cost_pct, self.sell_cost_pct = buy_cost_pct, sell_cost_pct
        self.reward_scaling, self.initial_capital = reward_scaling, initial_capital
        self.env_name = "StockEnv"                                                  # :60
        self.state_dim, self.action_dim = v.state_dim, v.action_dim                 # :63-65
        self.max_step = v.max_step                                                  # :67
        self.if_train = bool(config["if_train"])
        self.if_discrete = False
        self.target_return = 10.0
        self.episode_return = 0.0
        self.observation_space = Box(low=-3000, high=3000, shape=(self.state_dim,), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(self.action_dim,), dtype=np.float32)
        self._sync()

    @classmethod
    def make_vec(cls, config, num_envs, **kw):
        return VecStockTradingEnvNP(config, num_envs, **kw)

    def _sync(self):
        st = self._vec.state_numpy()
        self.day = int(st["day"][0])
        self.amount = st["amount"][0]
        self.stocks = st["stocks"][0]
        self.stocks_cool_down = st["cool_down"][0]
        self.total_asset = st["total_asset"][0]
        self.initial_total_asset = st["initial_total_asset"][0]
        self.gamma_reward = st["gamma_reward"][0]

    def reset(self):
        obs = self._vec.reset().cpu().numpy()[0]
        self._sync()
        return obs

    def step(self, actions):
        obs, rew, done, _ = self._vec.step(to_action_tensor(self._vec, actions))
        self._sync()
        d = bool(done.cpu().numpy()[0])
        if d:
            self.episode_return = float(self._vec.state_numpy()["episode_return"][0])   # :143-145
        return obs.cpu().numpy()[0], float(rew.cpu().numpy()[0]), d, dict()
