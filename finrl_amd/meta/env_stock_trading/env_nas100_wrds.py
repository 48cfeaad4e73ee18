"""Drop-in for ``finrl.meta.env_stock_trading.env_nas100_wrds.StockEnvNAS100``
(env_nas100_wrds.py:14-237 in the reference tree): the array-state stock env on minute-level
NASDAQ-100 arrays -- same ``step`` as ``env_stocktrading_np.StockTradingEnv`` (:115-158 vs
env_stocktrading_np.py:103-147), a random start state at EVERY ``reset()`` (:98-113) and an
observation whose first entry is ``max(amount, 1e4) * 2**-12`` (:160-161).  One HIP launch per step
through the C ABI (finenv_stocknp_*, ``obs_amount_floor = 1e4``).

Differences from the reference, all at the constructor:
  * arrays are used as float32 (what its ``load_data`` produces, :180-181); float64 arrays handed
    over with ``cwd=None`` are cast, where the reference would silently compute in float64;
  * with ``cwd`` set the reference builds a tuple of a tuple and fails at the first slice (:41-49);
    here the three ``.npy`` files are loaded as ``load_data`` describes and used.
``reset()`` draws from NumPy's global generator exactly as the reference does (same calls, same
order), so a seeded ``numpy.random`` reproduces the reference's start states."""
from __future__ import annotations

import os

import numpy as np
from numpy import random as rd

from ...vec_stocknp import TAG_F32, VecStockTradingEnvNP
from .._single import to_action_tensor


class StockEnvNAS100:
    def __init__(self, cwd="./data/nas100", price_ary=None, tech_ary=None, turbulence_ary=None,
                 gamma=0.999, turbulence_thresh=30, min_stock_rate=0.1, max_stock=1e2,
                 initial_capital=1e6, buy_cost_pct=1e-3, sell_cost_pct=1e-3, data_gap=4,
                 reward_scaling=2 ** -11, ticker_list=None, tech_indicator_list=None,
                 initial_stocks=None, if_eval=False, if_trade=False, device="cuda"):
        self.min_stock_rate = min_stock_rate
        beg_i, mid_i, end_i = 0, int(211210), int(422420)                          # :37
        (i0, i1) = (beg_i, mid_i) if if_eval else (mid_i, end_i)
        if cwd is not None:
            price_ary, tech_ary, turbulence_ary = self.load_data(cwd)
        arrays = [np.asarray(price_ary), np.asarray(tech_ary), np.asarray(turbulence_ary)]
        if not if_trade:                                                           # :46-51
            arrays = [a[i0:i1:data_gap] for a in arrays]
        else:
            arrays = [a[int(422420):int(528026):data_gap] for a in arrays]
        price, tech, turb = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        if price.shape[0] < 2:
            raise ValueError("StockEnvNAS100: the selected row range is empty (the index ranges "
                             ":37-51 are tuned to the 528,026-row WRDS minute data set)")
        stock_dim = price.shape[1]
        self.initial_stocks = (np.zeros(stock_dim, dtype=np.float32) if initial_stocks is None
                               else initial_stocks)                                # :65-69
        self._vec = VecStockTradingEnvNP(
            {"price_array": price, "tech_array": tech, "turbulence_array": turb, "if_train": False},
            1, gamma=gamma, turbulence_thresh=turbulence_thresh, min_stock_rate=min_stock_rate,
            max_stock=max_stock, initial_capital=initial_capital, buy_cost_pct=buy_cost_pct,
            sell_cost_pct=sell_cost_pct, reward_scaling=reward_scaling, auto_reset=False,
            device=device, obs_amount_floor=1e4)
        v = self._vec
        self.price_ary, self.tech_ary = v.price_ary, v.tech_ary
        self.turbulence_ary, self.turbulence_bool = v.turbulence_ary, v.turbulence_bool
        self.gamma, self.max_stock = gamma, max_stock
        self.buy_cost_pct, self.sell_cost_pct = buy_cost_pct, sell_cost_pct
        self.reward_scaling, self.initial_capital = reward_scaling, initial_capital
        self.day = self.amount = self.stocks = self.total_asset = None             # :72-77
        self.gamma_reward = self.initial_total_asset = self.stocks_cd = None
        self.env_name = "StockEnvNAS"                                              # :80
        self.state_dim = 1 + 2 + 3 * stock_dim + self.tech_ary.shape[1]            # :83
        self.action_dim = stock_dim
        self.max_step = self.price_ary.shape[0] - 1
        self.if_discrete = False
        self.target_return = 2.2
        self.episode_return = 0.0

    @classmethod
    def make_vec(cls, config, num_envs, **kw):
        """E device-resident envs stepped by one launch (set the start states with
        ``set_start_state`` or ``if_train=True`` in the config)."""
        kw.setdefault("obs_amount_floor", 1e4)
        kw.setdefault("gamma", 0.999)
        kw.setdefault("turbulence_thresh", 30)
        return VecStockTradingEnvNP(config, num_envs, **kw)

    def _sync(self):
        st = self._vec.state_numpy()
        self.day = int(st["day"][0])
        self.amount = st["amount"][0]
        self.stocks = st["stocks"][0]
        self.stocks_cd = st["cool_down"][0]
        self.total_asset = st["total_asset"][0]
        self.initial_total_asset = st["initial_total_asset"][0]
        self.gamma_reward = st["gamma_reward"][0]

    def reset(self):                                                               # :95-110
        price = self.price_ary[0]
        stocks = (self.initial_stocks + rd.randint(0, 64, size=self.initial_stocks.shape)
                  ).astype(np.float32)
        amount = self.initial_capital * rd.uniform(0.95, 1.05) - (stocks * price).sum()
        # (NumPy >= 2: python float - np.float32 -> np.float32, the value the reference holds)
        self._vec.set_start_state(stocks, float(np.float32(amount)), TAG_F32)
        obs = self._vec.reset().cpu().numpy()[0]
        self._sync()
        return obs

    def step(self, actions):                                                       # :112-158
        obs, rew, done, _ = self._vec.step(to_action_tensor(self._vec, actions))
        self._sync()
        d = bool(done.cpu().numpy()[0])
        if d:
            self.episode_return = float(self._vec.state_numpy()["episode_return"][0])
        return obs.cpu().numpy()[0], float(rew.cpu().numpy()[0]), d, dict()

    def load_data(self, cwd):                                                      # :173-189
        turbulence_ary = np.load(f"{cwd}/turb_ary.npy")
        turbulence_ary = turbulence_ary.repeat(390)[-528026:]
        price_ary = tech_ary = None
        if os.path.exists(f"{cwd}/price_ary.npy"):
            price_ary = np.load(f"{cwd}/price_ary.npy").astype(np.float32)
            tech_ary = np.load(f"{cwd}/tech_ary.npy").astype(np.float32)
        return price_ary, tech_ary, turbulence_ary

    def draw_cumulative_return(self, args, _torch) -> list:                        # :191-229
        agent = args.agent
        agent.init(args.net_dim, self.state_dim, self.action_dim)
        agent.save_load_model(cwd=args.cwd, if_save=False)
        act, device = agent.act, agent.device
        state = self.reset()
        episode_returns = []
        with _torch.no_grad():
            for _ in range(self.max_step):
                a_tensor = act(_torch.as_tensor((state,), device=device))
                state, reward, done, _ = self.step(a_tensor.detach().cpu().numpy()[0])
                total_asset = self.amount + (self.price_ary[self.day] * self.stocks).sum()
                episode_returns.append(total_asset / self.initial_total_asset)
                if done:
                    break
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.plot(episode_returns)
        plt.grid()
        plt.title("cumulative return")
        plt.xlabel("day")
        plt.xlabel("multiple of initial_account")
        plt.savefig(f"{args.cwd}/cumulative_return.jpg")
        print(f"| draw_cumulative_return: save in {args.cwd}/cumulative_return.jpg")
        return episode_returns

    @staticmethod
    def sigmoid_sign(ary, thresh):                                                 # :231-236
        def sigmoid(x):
            return 1 / (1 + np.exp(-x * np.e)) - 0.5
        return sigmoid(ary / thresh) * thresh
