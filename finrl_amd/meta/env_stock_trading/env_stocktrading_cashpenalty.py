"""Drop-in for ``finrl.meta.env_stock_trading.env_stocktrading_cashpenalty.StockTradingEnvCashpenalty``
(env_stocktrading_cashpenalty.py:19-409 in the reference tree): same constructor keywords,
``reset() / step()``, ``cash_on_hand`` / ``holdings`` / ``closings`` / ``current_step``
properties, ``get_sb_env``; one HIP launch per step through the C ABI (finenv_cashpenalty_*).
Observations are assembled in float64 from the device state and the host copy of the frame (the
reference returns float64 lists).  Not reproduced: the periodic console log (``print_verbosity``)
and the SB3 logger records (:181-215)."""
from __future__ import annotations

import random
import time

import numpy as np

from ...spaces import Box
from ...vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
from .._single import to_action_tensor


class StockTradingEnvCashpenalty:
    metadata = {"render.modes": ["human"]}
    _vec_cls = VecCashPenaltyEnv

    def __init__(self, df, buy_cost_pct=3e-3, sell_cost_pct=3e-3, date_col_name="date", hmax=10,
                 discrete_actions=False, shares_increment=1, turbulence_threshold=None,
                 print_verbosity=10, initial_amount=1e6,
                 daily_information_cols=["open", "close", "high", "low", "volume"],
                 cache_indicator_data=True, cash_penalty_proportion=0.1, random_start=True,
                 patient=False, currency="$", device="cuda", **extra):
        self.df = df
        self.panel = df if isinstance(df, CashPenaltyPanel) else \
            CashPenaltyPanel.from_dataframe(df, daily_information_cols, date_col_name)
        self.assets, self.dates = self.panel.assets, self.panel.dates
        self.random_start, self.discrete_actions, self.patient = random_start, discrete_actions, patient
        self.currency, self.shares_increment, self.hmax = currency, shares_increment, hmax
        self.initial_amount, self.print_verbosity = initial_amount, print_verbosity
        self.buy_cost_pct, self.sell_cost_pct = buy_cost_pct, sell_cost_pct
        self.turbulence_threshold = turbulence_threshold
        self.daily_information_cols = list(daily_information_cols)
        self.cash_penalty_proportion = cash_penalty_proportion
        self.state_space = 1 + len(self.assets) + len(self.assets) * len(self.daily_information_cols)
        self.action_space = Box(low=-1, high=1, shape=(len(self.assets),))
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.state_space,))
        self.turbulence = 0
        self.episode = -1                                                           # :98
        self.episode_history = []
        self._vec = self._vec_cls(
            self.panel, 1, buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct, hmax=hmax,
            discrete_actions=discrete_actions, shares_increment=shares_increment,
            turbulence_threshold=turbulence_threshold, initial_amount=initial_amount,
            cash_penalty_proportion=cash_penalty_proportion, random_start=False, patient=patient,
            auto_reset=False, device=device, **extra)
        self.date_index = self.starting_point = 0
        self._st = None

    @classmethod
    def make_vec(cls, df, num_envs, daily_information_cols=("open", "close", "high", "low", "volume"),
                 date_col_name="date", **kw):
        panel = df if isinstance(df, CashPenaltyPanel) else \
            CashPenaltyPanel.from_dataframe(df, list(daily_information_cols), date_col_name)
        return cls._vec_cls(panel, num_envs, **kw)

    def seed(self, seed=None):                                                      # :121-124
        if seed is None:
            seed = int(round(time.time() * 1000))
        random.seed(seed)

    @property
    def current_step(self):
        return self.date_index - self.starting_point

    @property
    def cash_on_hand(self):
        return float(self._st["coh"][0])

    @property
    def holdings(self):
        return self._st["holdings"][0]

    @property
    def closings(self):
        return np.array(self.panel.close[self.date_index])

    def _sync(self):
        self._st = self._vec.state_numpy()
        self.date_index = int(self._st["date_index"][0])
        self.starting_point = int(self._st["start"][0])
        self.episode = int(self._st["episode"][0])
        self.sum_trades = float(self._st["sum_trades"][0])
        self.turbulence = float(self._st["turbulence"][0])
        return np.concatenate([[self._st["coh"][0]], self._st["holdings"][0],
                               self.panel.info[self.date_index].reshape(-1)])

    def reset(self):                                                                # :131-157
        self.seed()
        start = random.choice(range(int(len(self.dates) * 0.5))) if self.random_start else 0
        self._vec.set_next_start(start)
        self._vec.reset()
        return self._sync()

    def step(self, actions):                                                        # :291-372
        _, rew, done, _ = self._vec.step(to_action_tensor(self._vec, actions))
        state = self._sync()
        return state, float(rew.cpu().numpy()[0]), bool(done.cpu().numpy()[0]), {}

    def get_sb_env(self):
        from ...vec_env import SingleEnvVecAdapter
        e = SingleEnvVecAdapter(self)
        return e, e.reset()
