"""Drop-in for ``finrl.meta.env_portfolio_allocation.env_portfolio.StockPortfolioEnv``
(env_portfolio.py:14-261 in the reference tree): same constructor keywords, ``reset() / step()``,
``softmax_normalization``, ``save_asset_memory`` / ``save_action_memory`` and ``get_sb_env``; one
HIP launch per step through the C ABI (finenv_portfolio_*).  ``df`` carries a ``cov_list`` column
(``finrl_amd.riskpre.add_cov_list`` builds it on the GPU).  The reference writes two PNG plots at
the end of every episode (:131-139); this facade does not."""
from __future__ import annotations

import numpy as np

from ...panel import PortfolioPanel
from ...spaces import Box
from ...vec_portfolio import VecStockPortfolioEnv
from .._single import to_action_tensor


class StockPortfolioEnv:
    metadata = {"render.modes": ["human"]}

    def __init__(self, df, stock_dim, hmax, initial_amount, transaction_cost_pct, reward_scaling,
                 state_space, action_space, tech_indicator_list, turbulence_threshold=None,
                 lookback=252, day=0, device="cuda"):
        self.day, self.lookback, self.df = day, lookback, df
        self.stock_dim, self.hmax, self.initial_amount = stock_dim, hmax, initial_amount
        self.transaction_cost_pct, self.reward_scaling = transaction_cost_pct, reward_scaling
        self.state_space = state_space
        self.tech_indicator_list = list(tech_indicator_list)
        self.action_space = Box(low=0, high=1, shape=(action_space,))               # :88
        self.observation_space = Box(low=-np.inf, high=np.inf,
                                     shape=(state_space + len(self.tech_indicator_list),
                                            state_space))                           # :91-95
        self.turbulence_threshold = turbulence_threshold
        self.panel = df if isinstance(df, PortfolioPanel) else \
            PortfolioPanel.from_dataframe(df, self.tech_indicator_list)
        self._vec = VecStockPortfolioEnv(self.panel, 1, initial_amount=initial_amount,
                                         auto_reset=False, device=device)
        self._vec.enable_weights()
        self._begin_episode()

    @classmethod
    def make_vec(cls, df, num_envs, tech_indicator_list, **kw):
        panel = df if isinstance(df, PortfolioPanel) else \
            PortfolioPanel.from_dataframe(df, list(tech_indicator_list))
        return VecStockPortfolioEnv(panel, num_envs, **kw)

    def _state_at(self, t):                       # np.append(covs, tech rows, axis=0), :172-179
        return np.append(np.asarray(self.panel.cov[t], dtype=np.float64),
                         np.asarray(self.panel.tech[t], dtype=np.float64), axis=0)

    def _begin_episode(self):
        self.day = 0
        self.terminal = False
        self.portfolio_value = self.initial_amount
        self.asset_memory = [self.initial_amount]                                   # :114-120
        self.portfolio_return_memory = [0]
        self.actions_memory = [[1 / self.stock_dim] * self.stock_dim]
        self.date_memory = [self.panel.dates[0]]
        self.state = self._state_at(0)
        self.reward = self.initial_amount

    def reset(self):                                                                # :202-220
        self._vec.reset()
        self._begin_episode()
        return self.state

    def softmax_normalization(self, actions):                                       # :225-229
        numerator = np.exp(actions)
        return numerator / np.sum(np.exp(actions))

    def step(self, actions):                                                        # :125-200
        self.terminal = self.day >= self.panel.T - 1
        obs, rew, done, _ = self._vec.step(to_action_tensor(self._vec, actions))
        st = self._vec.state_numpy()
        if self.terminal:
            return self.state, self.reward, self.terminal, {}
        self.day = int(st["day"][0])
        new_value = float(st["value"][0])
        self.state = self._state_at(self.day)
        self.actions_memory.append(self._vec.weights.cpu().numpy()[0].astype(np.float64))
        self.portfolio_return_memory.append(new_value / self.portfolio_value - 1)
        self.portfolio_value = new_value
        self.date_memory.append(self.panel.dates[self.day])
        self.asset_memory.append(new_value)
        self.reward = new_value                                                      # :196-198
        return self.state, self.reward, self.terminal, {}

    def render(self, mode="human"):
        return self.state

    def save_asset_memory(self):                                                    # :231-238
        import pandas as pd
        return pd.DataFrame({"date": self.date_memory, "daily_return": self.portfolio_return_memory})

    def save_action_memory(self):                                                   # :240-250
        import pandas as pd
        df_actions = pd.DataFrame(self.actions_memory)
        df_actions.columns = list(self.panel.tickers)
        df_actions.index = pd.DataFrame({"date": self.date_memory}).date
        return df_actions

    def get_sb_env(self):                                                           # :256-259
        from ...vec_env import SingleEnvVecAdapter
        e = SingleEnvVecAdapter(self)
        return e, e.reset()
