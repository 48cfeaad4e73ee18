"""Drop-in for the reference's ``finrl.meta.env_stock_trading.env_stocktrading.StockTradingEnv``
(env_stocktrading.py:19-552 in the reference tree): same constructor keywords, same
gym-0.21 protocol (``reset() -> obs``, ``step(a) -> (obs, reward, done, info)``), same
harness surface (``save_asset_memory``, ``save_action_memory``, ``render``, ``get_sb_env``),
but every step is one launch of the HIP kernel through the C ABI.

This single-env facade exists for API compatibility (back-tests, ``DRL_prediction``,
notebooks).  It wraps a 1-env :class:`finrl_amd.vec_env.VecStockTradingEnv`; after each launch
it copies the handful of per-env scalars back and assembles the observation in float64 from
the host copy of the panel, so the returned numbers are the same doubles the reference keeps
in its Python list state.  Throughput work should use ``make_vec(...)`` /
``VecStockTradingEnv`` directly (tens of thousands of envs per launch, no host round trip).
"""
from __future__ import annotations

import os

import numpy as np

from ...panel import StockPanel
from ...spaces import Box
from ...vec_env import SingleEnvVecAdapter, VecStockTradingEnv


class StockTradingEnv:
    """A stock trading environment (MI355X-native engine behind the reference's interface)."""

    metadata = {"render.modes": ["human"]}

    def __init__(self, df, stock_dim, hmax, initial_amount, num_stock_shares, buy_cost_pct,
                 sell_cost_pct, reward_scaling, state_space, action_space, tech_indicator_list,
                 turbulence_threshold=None, risk_indicator_col="turbulence", make_plots=False,
                 print_verbosity=10, day=0, initial=True, previous_state=[], model_name="",
                 mode="", iteration="", device="cuda", reset_quirk=True):
        self.day = day
        self.df = df
        self.stock_dim = stock_dim
        self.hmax = hmax
        self.num_stock_shares = list(num_stock_shares)
        self.initial_amount = initial_amount
        self.buy_cost_pct = buy_cost_pct
        self.sell_cost_pct = sell_cost_pct
        self.reward_scaling = reward_scaling
        self.state_space = state_space
        self.tech_indicator_list = list(tech_indicator_list)
        self.action_space = Box(low=-1, high=1, shape=(action_space,))          # :60
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(state_space,))  # :61-63
        self.terminal = False
        self.make_plots = make_plots
        self.print_verbosity = print_verbosity
        self.turbulence_threshold = turbulence_threshold
        self.risk_indicator_col = risk_indicator_col
        self.initial = initial
        self.previous_state = previous_state
        self.model_name = model_name
        self.mode = mode
        self.iteration = iteration

        self.panel = df if isinstance(df, StockPanel) else StockPanel.from_dataframe(
            df, self.tech_indicator_list, risk_indicator_col)
        if self.panel.N != stock_dim:
            raise ValueError(f"stock_dim={stock_dim} but the frame holds {self.panel.N} tickers")
        if state_space != self.panel.D:
            raise ValueError(f"state_space={state_space}, expected 1+2N+KN={self.panel.D}")
        if initial:
            cash0, shares0 = initial_amount, self.num_stock_shares
        else:   # carry-over from a previous window, :423-440
            cash0 = previous_state[0]
            shares0 = [int(x) for x in previous_state[stock_dim + 1:2 * stock_dim + 1]]
        self._vec = VecStockTradingEnv(
            self.panel, 1, hmax=hmax, initial_amount=cash0, num_stock_shares=shares0,
            buy_cost_pct=buy_cost_pct, sell_cost_pct=sell_cost_pct,
            reward_scaling=reward_scaling, turbulence_threshold=turbulence_threshold, day=day,
            initial=initial, reset_quirk=reset_quirk, track_stats=True, auto_reset=False,
            device=device)
        self._vec.enable_realised()
        self._torch = __import__("torch")

        self.reward = 0
        self.turbulence = 0
        self.cost = 0
        self.trades = 0
        self.episode = 0
        self._pull()
        self.asset_memory = [self._asset0]                                   # :85-91
        self.rewards_memory = []
        self.actions_memory = []
        self.state_memory = []
        self.date_memory = [self._get_date()]
        self._seed()

    # ------------------------------------------------------------------ device <-> host
    def _pull(self):
        """Copy the env's scalars back and rebuild the float64 list state (:453-478)."""
        st = self._vec.state_numpy()
        self._cash = float(st["cash"][0])
        self._shares = st["shares"][0].astype(np.int64)
        self.day = int(st["day"][0])
        self._price_day = int(st["price_day"][0])
        self.cost = float(st["cost"][0])
        self.trades = int(st["trades"][0])
        self.turbulence = float(st["turbulence"][0]) if self.turbulence_threshold is not None \
            else 0
        self._asset0 = float(st["asset0"][0])
        self._last_reward = float(st["last_reward"][0])
        self._begin_asset = float(st["begin_asset"][0])   # begin_total_asset of the NEXT step (:311-314)
        row = self._price_day
        self.state = ([self._cash] + self.panel.close[row].tolist() + self._shares.tolist()
                      + self.panel.tech[row].reshape(-1).tolist())
        return self.state

    def _total_asset(self):
        p = np.asarray(self.state[1:self.stock_dim + 1])
        h = np.asarray(self.state[self.stock_dim + 1:2 * self.stock_dim + 1])
        return self.state[0] + sum(p * h)

    # ------------------------------------------------------------------ gym protocol
    def _device_actions(self, actions):
        """float32 actions go to the kernel as they are (it scales and truncates in float32 like
        :304-305 does for SB3's float32 arrays).  Any other dtype (float64 arrays, Python lists) is
        scaled HERE in the caller's dtype -- exactly the reference's `(actions * hmax).astype(int)`
        -- and the integer share counts are handed over as float32 values k' = (k +- 0.5) / hmax
        that the kernel's float32 multiply-truncate maps back to k exactly (checked below)."""
        a = np.asarray(actions)
        if a.dtype == np.float32:
            return a.reshape(1, -1)
        k = (a * self.hmax).astype(int).reshape(1, -1)                          # :304-305
        enc = ((k + 0.5 * np.sign(k)) / self.hmax).astype(np.float32) if self.hmax else \
            np.zeros(k.shape, np.float32)
        if not np.array_equal((enc * np.float32(self.hmax)).astype(np.int64), k):
            raise ValueError("non-float32 actions: scaled share counts exceed the exactly "
                             "representable range; pass float32 actions")
        return enc

    def step(self, actions):
        torch = self._torch
        a = torch.as_tensor(np.ascontiguousarray(self._device_actions(actions)))
        was_terminal_day = self.day >= self.panel.T - 1
        begin_total_asset = self._begin_asset
        _, rew, done, _ = self._vec.step(a.to(self._vec.device))
        self.terminal = bool(done.cpu().numpy()[0])
        self._pull()
        if self.terminal:
            self._terminal_branch()
            return self.state, self.reward, self.terminal, {}
        assert not was_terminal_day
        self.actions_memory.append(self._vec.realised[0].cpu().numpy().astype(np.int64))
        end_total_asset = self._total_asset()
        self.asset_memory.append(end_total_asset)
        self.date_memory.append(self._get_date())
        self.reward = self._last_reward                     # (end - begin) * reward_scaling
        self.rewards_memory.append(end_total_asset - begin_total_asset)      # unscaled, :350-351
        self.state_memory.append(self.state)
        return self.state, self.reward, self.terminal, {}

    def _terminal_branch(self):
        """:222-301 -- summary print, optional CSV / PNG dumps."""
        import pandas as pd
        if self.make_plots:
            self._make_plot()
        stats = self._vec.episode_stats().cpu().numpy()[0]
        end_total_asset = self._total_asset()
        tot_reward = end_total_asset - self.asset_memory[0]
        sharpe = stats[5]
        if self.episode % self.print_verbosity == 0:
            print(f"day: {self.day}, episode: {self.episode}")
            print(f"begin_total_asset: {self.asset_memory[0]:0.2f}")
            print(f"end_total_asset: {end_total_asset:0.2f}")
            print(f"total_reward: {tot_reward:0.2f}")
            print(f"total_cost: {self.cost:0.2f}")
            print(f"total_trades: {self.trades}")
            if not np.isnan(sharpe):
                print(f"Sharpe: {sharpe:0.3f}")
            print("=================================")
        if self.model_name != "" and self.mode != "":
            os.makedirs("results", exist_ok=True)
            tag = f"{self.mode}_{self.model_name}_{self.iteration}"
            self.save_action_memory().to_csv(f"results/actions_{tag}.csv")
            df_total_value = pd.DataFrame({"account_value": self.asset_memory})
            df_total_value["date"] = self.date_memory
            df_total_value["daily_return"] = df_total_value["account_value"].pct_change(1)
            df_total_value.to_csv(f"results/account_value_{tag}.csv", index=False)
            df_rewards = pd.DataFrame({"account_rewards": self.rewards_memory})
            df_rewards["date"] = self.date_memory[:-1]
            df_rewards.to_csv(f"results/account_rewards_{tag}.csv", index=False)
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                plt.plot(self.asset_memory, "r")
                plt.savefig(f"results/account_value_{tag}.png")
                plt.close()
            except ImportError:
                pass

    def reset(self):
        self._vec.reset()
        self._pull()
        self.asset_memory = [self._asset0]                                   # :364-378
        self.turbulence = 0
        self.cost = 0
        self.trades = 0
        self.terminal = False
        self.rewards_memory = []
        self.actions_memory = []
        self.date_memory = [self.panel.dates[0]]                             # :389 (day 0)
        self.episode += 1
        return self.state

    def render(self, mode="human", close=False):
        return self.state

    # ------------------------------------------------------------------ harness surface
    def _make_plot(self):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        os.makedirs("results", exist_ok=True)
        plt.plot(self.asset_memory, "r")
        plt.savefig(f"results/account_value_trade_{self.episode}.png")
        plt.close()

    def _get_date(self):
        return self.panel.dates[self.day]

    def save_asset_memory(self):
        import pandas as pd
        return pd.DataFrame({"date": self.date_memory, "account_value": self.asset_memory})

    def save_action_memory(self):
        import pandas as pd
        date_list = self.date_memory[:-1]
        if self.stock_dim > 1:
            df_actions = pd.DataFrame(self.actions_memory)
            df_actions.columns = self.panel.tickers
            df_actions.index = pd.Index(date_list, name="date")
        else:
            df_actions = pd.DataFrame({"date": date_list, "actions": self.actions_memory})
        return df_actions

    def save_state_memory(self):
        import pandas as pd
        return pd.DataFrame({"date": self.date_memory[:-1], "states": self.state_memory})

    def _seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def get_sb_env(self):
        """(vec_env, first_obs) as the reference returns from DummyVecEnv([lambda: self]), :549-552.
        With stable-baselines3 installed its DummyVecEnv wraps this object; otherwise the built-in
        DummyVecEnv-shaped adapter wraps it -- either way every step goes through this facade, so
        ``asset_memory`` / ``actions_memory`` keep filling and ``env_method("save_asset_memory")``
        (agents/stablebaselines3/models.py:120-121) finds its target."""
        try:
            from stable_baselines3.common.vec_env import DummyVecEnv
            e = DummyVecEnv([lambda: self])
        except ImportError:
            e = SingleEnvVecAdapter(self)
        obs = e.reset()
        return e, obs

    # ------------------------------------------------------------------ batched constructor
    @classmethod
    def make_vec(cls, df, num_envs, *, tech_indicator_list, hmax, initial_amount,
                 num_stock_shares=None, buy_cost_pct=1e-3, sell_cost_pct=1e-3,
                 reward_scaling=1e-4, turbulence_threshold=None,
                 risk_indicator_col="turbulence", device="cuda", **kw):
        """E device-resident copies of this env stepped by one launch (VecStockTradingEnv)."""
        panel = df if isinstance(df, StockPanel) else StockPanel.from_dataframe(
            df, tech_indicator_list, risk_indicator_col)
        return VecStockTradingEnv(panel, num_envs, hmax=hmax, initial_amount=initial_amount,
                                  num_stock_shares=num_stock_shares, buy_cost_pct=buy_cost_pct,
                                  sell_cost_pct=sell_cost_pct, reward_scaling=reward_scaling,
                                  turbulence_threshold=turbulence_threshold, device=device, **kw)
