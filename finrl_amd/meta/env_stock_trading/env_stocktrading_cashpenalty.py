"""Drop-in for ``finrl.meta.env_stock_trading.env_stocktrading_cashpenalty.StockTradingEnvCashpenalty``
(env_stocktrading_cashpenalty.py:19-409 in the reference tree): same constructor keywords,
``reset() / step()``, ``cash_on_hand`` / ``holdings`` / ``closings`` / ``current_step``
properties and the harness surface ``DRLAgent.DRL_prediction`` drives
(``get_sb_env``, ``get_multiproc_env``, ``save_asset_memory``, ``save_action_memory``,
``account_information``, ``actions_memory``, ``transaction_memory``, ``state_memory``,
``episode_history`` and the console log).

Every step is one HIP launch through the C ABI (finenv_cashpenalty_*).  The harness lists are a
host-side LOG of what the kernel did: per step the kernel writes one float64 "audit" row (begin
cash, asset value, reward, reason flags, applied transactions; ``finenv_cashpenalty_set_audit``)
which this facade appends to the lists the reference keeps -- no trading arithmetic happens here.
Observations are assembled in float64 from the device state and the host copy of the frame (the
reference returns float64 lists)."""
from __future__ import annotations

import contextlib
import io
import random
import time

import numpy as np

from ... import _native as nat
from ...spaces import Box
from ...vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
from .._single import to_action_tensor


def _sb3_logger():
    """``stable_baselines3.common.logger`` when it is installed (:12), else None."""
    try:
        from stable_baselines3.common import logger
        return logger
    except ImportError:
        return None


class StockTradingEnvCashpenalty:
    metadata = {"render.modes": ["human"]}
    _vec_cls = VecCashPenaltyEnv
    _row_fmt = "{0:4}|{1:4}|{2:15}|{3:15}|{4:15}|{5:10}|{6:10}|{7:10}"            # :229
    _header = ("EPISODE", "STEPS", "TERMINAL_REASON", "CASH", "TOT_ASSETS",
               "TERMINAL_REWARD_unsc", "GAINLOSS_PCT", "CASH_PROPORTION")          # :232-241

    def __init__(self, df, buy_cost_pct=3e-3, sell_cost_pct=3e-3, date_col_name="date", hmax=10,
                 discrete_actions=False, shares_increment=1, turbulence_threshold=None,
                 print_verbosity=10, initial_amount=1e6,
                 daily_information_cols=["open", "close", "high", "low", "volume"],
                 cache_indicator_data=True, cash_penalty_proportion=0.1, random_start=True,
                 patient=False, currency="$", device="cuda", **extra):
        self._ctor = dict(buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct,
                          date_col_name=date_col_name, hmax=hmax,
                          discrete_actions=discrete_actions, shares_increment=shares_increment,
                          turbulence_threshold=turbulence_threshold,
                          print_verbosity=print_verbosity, initial_amount=initial_amount,
                          daily_information_cols=list(daily_information_cols),
                          cache_indicator_data=cache_indicator_data,
                          cash_penalty_proportion=cash_penalty_proportion,
                          random_start=random_start, patient=patient, currency=currency,
                          device=device)
        self._extra = dict(extra)
        self.panel = df if isinstance(df, CashPenaltyPanel) else \
            CashPenaltyPanel.from_dataframe(df, daily_information_cols, date_col_name)
        # callers read `environment.df.index.unique()` (agents/stablebaselines3/models.py:117):
        # the reference keeps the frame indexed by date (:75)
        self.df = df.set_index(date_col_name) if hasattr(df, "set_index") else df
        self.stock_col = "tic"
        self.assets, self.dates = self.panel.assets, self.panel.dates
        self.random_start, self.discrete_actions, self.patient = random_start, discrete_actions, patient
        self.currency, self.shares_increment, self.hmax = currency, shares_increment, hmax
        self.initial_amount, self.print_verbosity = initial_amount, print_verbosity
        self.buy_cost_pct, self.sell_cost_pct = buy_cost_pct, sell_cost_pct
        self.turbulence_threshold = turbulence_threshold
        self.daily_information_cols = list(daily_information_cols)
        self.cash_penalty_proportion = cash_penalty_proportion
        self.cache_indicator_data = cache_indicator_data
        if cache_indicator_data:                 # :103-108 (the packed panel IS the cache here)
            print("caching data")
            print("data cached!")
        self.state_space = 1 + len(self.assets) + len(self.assets) * len(self.daily_information_cols)
        self.action_space = Box(low=-1, high=1, shape=(len(self.assets),))
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.state_space,))
        self.turbulence = 0
        self.episode = -1                                                           # :98
        self.episode_history = []
        self.printed_header = False
        self._vec = self._vec_cls(
            self.panel, 1, buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct, hmax=hmax,
            discrete_actions=discrete_actions, shares_increment=shares_increment,
            turbulence_threshold=turbulence_threshold, initial_amount=initial_amount,
            cash_penalty_proportion=cash_penalty_proportion, random_start=False, patient=patient,
            auto_reset=False, device=device, **extra)
        self._audit = self._vec.enable_audit()
        self.date_index = self.starting_point = 0
        self._st = None
        self.sum_trades = 0
        self.actions_memory, self.transaction_memory, self.state_memory = [], [], []
        self.account_information = {"cash": [], "asset_value": [], "total_assets": [], "reward": []}

    # ------------------------------------------------------------------ batched constructor
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

    # ------------------------------------------------------------------ properties (:113-130)
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

    def get_date_vector(self, date, cols=None):                                     # :159-171
        if cols is None:
            return self.panel.info[date].reshape(-1).tolist()
        out = []
        for j in range(len(self.assets)):
            for c in cols:
                if c == "close":
                    out.append(float(self.panel.close[date, j]))
                elif c == "turbulence":
                    out.append(float(self.panel.turb[date]))
                else:
                    out.append(float(self.panel.info[date, j, self.daily_information_cols.index(c)]))
        return out

    # ------------------------------------------------------------------ device <-> host
    def _sync(self):
        self._st = self._vec.state_numpy()
        self.date_index = int(self._st["date_index"][0])
        self.starting_point = int(self._st["start"][0])
        self.episode = int(self._st["episode"][0])
        self.sum_trades = float(self._st["sum_trades"][0])
        self.turbulence = float(self._st["turbulence"][0])
        return np.concatenate([[self._st["coh"][0]], self._st["holdings"][0],
                               self.panel.info[self.date_index].reshape(-1)])

    def _audit_row(self):
        return self._audit.cpu().numpy()[0]

    # ------------------------------------------------------------------ gym protocol
    def reset(self):                                                                # :131-157
        self.seed()
        start = random.choice(range(int(len(self.dates) * 0.5))) if self.random_start else 0
        self._vec.set_next_start(start)
        self._vec.reset()
        init_state = self._sync()
        self.sum_trades = 0
        self.actions_memory = []
        self.transaction_memory = []
        self.state_memory = [init_state]
        self.account_information = {"cash": [], "asset_value": [], "total_assets": [], "reward": []}
        return init_state

    def _record_action(self, actions, closings):
        """what the reference appends to actions_memory (:264: the raw action array)."""
        self.actions_memory.append(actions)

    def _reasons_before_shortage(self, flags):
        return [("TURBULENCE", flags & nat.AUDIT_F_TURBULENCE)]                     # :282-287

    def _reasons_after_trades(self, flags):
        return []

    def step(self, actions):                                                        # :291-372
        actions = np.asarray(actions)
        self.log_header()
        if (self.current_step + 1) % self.print_verbosity == 0:                     # :296-297
            self.log_step(reason="update")
        closings = self.closings
        _, _, done, _ = self._vec.step(to_action_tensor(self._vec, actions))
        au = self._audit_row()
        flags = int(au[nat.AUDIT_HEAD - 1])
        reward = float(au[2])
        done = bool(done.cpu().numpy()[0])
        if flags & nat.AUDIT_F_LAST_DATE:                                           # :299-301
            self._sync()
            return self.return_terminal(reward=reward)
        self.account_information["cash"].append(float(au[0]))                       # :312-315
        self.account_information["asset_value"].append(float(au[1]))
        self.account_information["total_assets"].append(float(au[0]) + float(au[1]))
        self.account_information["reward"].append(reward)                           # :318
        self._record_action(actions, closings)
        for reason, hit in self._reasons_before_shortage(flags):
            if hit:
                self.log_step(reason=reason)
        if flags & nat.AUDIT_F_CASH_SHORTAGE:                                       # :333-344
            if not self.patient:
                self._sync()
                return self.return_terminal(reason="CASH SHORTAGE", reward=reward)
            self.log_step(reason="CASH SHORTAGE")
        self.transaction_memory.append(np.array(au[nat.AUDIT_HEAD:]))               # :345-347
        for reason, hit in self._reasons_after_trades(flags):
            if hit:
                self.log_step(reason=reason)
        state = self._sync()
        self.state_memory.append(state)
        assert not done
        return state, reward, False, {}

    # ------------------------------------------------------------------ console / SB3 log
    def return_terminal(self, reason="Last Date", reward=0):                        # :173-205
        state = self.state_memory[-1]
        self.log_step(reason=reason, terminal_reward=reward)
        logger = _sb3_logger()
        if logger is not None:
            ai = self.account_information
            gl_pct = ai["total_assets"][-1] / self.initial_amount
            logger.record("environment/GainLoss_pct", (gl_pct - 1) * 100)
            logger.record("environment/total_assets", int(ai["total_assets"][-1]))
            logger.record("environment/total_reward_pct", (gl_pct - 1) * 100)
            logger.record("environment/total_trades", self.sum_trades)
            self._record_extra(logger)
            logger.record("environment/avg_daily_trades", self.sum_trades / self.current_step)
            logger.record("environment/avg_daily_trades_per_asset",
                          self.sum_trades / self.current_step / len(self.assets))
            logger.record("environment/completed_steps", self.current_step)
            logger.record("environment/sum_rewards", np.sum(ai["reward"]))
            logger.record("environment/cash_proportion", ai["cash"][-1] / ai["total_assets"][-1])
        return state, reward, True, {}

    def _record_extra(self, logger):
        pass

    def log_step(self, reason, terminal_reward=None):                               # :207-226
        ai = self.account_information
        if terminal_reward is None:
            terminal_reward = ai["reward"][-1]
        cash_pct = ai["cash"][-1] / ai["total_assets"][-1]
        gl_pct = ai["total_assets"][-1] / self.initial_amount
        rec = [self.episode, self.date_index - self.starting_point, reason,
               f"{self.currency}{'{:0,.0f}'.format(float(ai['cash'][-1]))}",
               f"{self.currency}{'{:0,.0f}'.format(float(ai['total_assets'][-1]))}",
               f"{terminal_reward*100:0.5f}%", f"{(gl_pct - 1)*100:0.5f}%",
               f"{cash_pct*100:0.2f}%"]
        self.episode_history.append(rec)
        print(self._row_fmt.format(*rec))

    def log_header(self):                                                           # :228-245
        if self.printed_header is False:
            self.template = self._row_fmt
            print(self._row_fmt.format(*self._header))
            self.printed_header = True

    # ------------------------------------------------------------------ harness surface
    def _clone(self):
        """A second env with this one's configuration AND current state (what ``deepcopy(self)``
        gives the reference inside get_sb_env, :374-376): fresh device handle, state copied."""
        kw = dict(self._ctor)
        kw.update(self._extra_ctor())
        with contextlib.redirect_stdout(io.StringIO()):      # deepcopy prints nothing
            other = type(self)(self.panel, **kw)
        other.df = self.df
        other._vec._f64.copy_(self._vec._f64)
        other._vec._i32.copy_(self._vec._i32)
        for k in ("episode", "episode_history", "printed_header", "sum_trades", "turbulence"):
            setattr(other, k, getattr(self, k) if k != "episode_history" else list(self.episode_history))
        other.actions_memory = list(self.actions_memory)
        other.transaction_memory = list(self.transaction_memory)
        other.state_memory = list(self.state_memory)
        other.account_information = {k: list(v) for k, v in self.account_information.items()}
        other._st = other._vec.state_numpy()
        other.date_index, other.starting_point = self.date_index, self.starting_point
        return other

    def _extra_ctor(self):
        return {}

    def __deepcopy__(self, memo):
        return self._clone()

    def get_sb_env(self):                                                           # :374-380
        from ...vec_env import SingleEnvVecAdapter
        try:
            from stable_baselines3.common.vec_env import DummyVecEnv
            e = DummyVecEnv([self._clone])
        except ImportError:
            e = SingleEnvVecAdapter(self._clone())
        obs = e.reset()
        return e, obs

    def get_multiproc_env(self, n=10):                                              # :382-388
        """The reference forks n copies into a SubprocVecEnv; here the n envs are ONE device batch
        stepped by one launch, behind the same VecEnv protocol."""
        kw = {k: self._ctor[k] for k in ("buy_cost_pct", "sell_cost_pct", "hmax", "discrete_actions",
                                         "shares_increment", "turbulence_threshold",
                                         "initial_amount", "cash_penalty_proportion",
                                         "random_start", "patient", "device")}
        kw.update(self._extra)
        e = self._vec_cls(self.panel, int(n), **kw).as_sb3_vec_env()
        obs = e.reset()
        return e, obs

    def save_asset_memory(self):                                                    # :390-397
        import pandas as pd
        if self.current_step == 0:
            return None
        self.account_information["date"] = self.dates[-len(self.account_information["cash"]):]
        return pd.DataFrame(self.account_information)

    def save_action_memory(self):                                                   # :399-409
        import pandas as pd
        if self.current_step == 0:
            return None
        return pd.DataFrame({"date": self.dates[-len(self.account_information["cash"]):],
                             "actions": self.actions_memory,
                             "transactions": self.transaction_memory})
