"""ctypes wrapper over oracle/stocknp_oracle.c -- TEST INFRASTRUCTURE (see stock.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .stock import lib, _p

TAG_PY, TAG_F32, TAG_F64 = 0, 1, 2
TAG_OF = {float: TAG_PY, np.float32: TAG_F32, np.float64: TAG_F64}


class NpCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_techw", C.c_int32),
                ("n_days", C.c_int32), ("max_stock_i", C.c_int32), ("min_action", C.c_int32),
                ("max_stock", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("reward_scaling", C.c_double),
                ("gamma", C.c_double), ("initial_capital", C.c_double),
                ("obs_amount_floor", C.c_double)]


def derive_arrays(price_array, tech_array, turbulence_array, turbulence_thresh=99):
    """env_stocktrading_np.py:27-35, 164-169 evaluated with the same NumPy expressions."""
    price = np.asarray(price_array).astype(np.float32)
    tech = np.asarray(tech_array).astype(np.float32)
    tech = tech * 2 ** -7
    turb = np.asarray(turbulence_array)
    turb_bool = (turb > turbulence_thresh).astype(np.float32)

    def sigmoid_sign(ary, thresh):
        def sigmoid(x):
            return 1 / (1 + np.exp(-x * np.e)) - 0.5
        return sigmoid(ary / thresh) * thresh
    turb_ary = (sigmoid_sign(turb, turbulence_thresh) * 2 ** -5).astype(np.float32)
    return (np.ascontiguousarray(price), np.ascontiguousarray(tech),
            np.ascontiguousarray(turb_ary), np.ascontiguousarray(turb_bool))


class StockNpOracle:
    def __init__(self, price_array, tech_array, turbulence_array, *, n_envs=1, gamma=0.99,
                 turbulence_thresh=99, min_stock_rate=0.1, max_stock=1e2, initial_capital=1e6,
                 buy_cost_pct=1e-3, sell_cost_pct=1e-3, reward_scaling=2 ** -11,
                 obs_amount_floor=0.0):
        self.price, self.tech, self.turb, self.turb_bool = derive_arrays(
            price_array, tech_array, turbulence_array, turbulence_thresh)
        T, N = self.price.shape
        self.E, self.N, self.W, self.T = int(n_envs), N, self.tech.shape[1], T
        self.D = 3 + 3 * N + self.W
        self.max_step = T - 1
        L = lib()
        L.np_oracle_create.restype = C.c_void_p
        self.cfg = NpCfg(self.E, N, self.W, T, int(max_stock), int(max_stock * min_stock_rate),
                         float(max_stock), float(buy_cost_pct), float(sell_cost_pct),
                         float(reward_scaling), float(gamma), float(initial_capital),
                         float(obs_amount_floor))
        self._h = C.c_void_p(L.np_oracle_create(C.byref(self.cfg), _p(self.price), _p(self.tech),
                                                _p(self.turb), _p(self.turb_bool)))

    def __del__(self):
        try:
            if self._h:
                lib().np_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_initial(self, stocks0, amount0, amount0_tag):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(stocks0, np.float32), (self.E, self.N)))
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(amount0, np.float64), (self.E,)))
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(amount0_tag, np.int32), (self.E,)))
        lib().np_oracle_set_initial(self._h, _p(s), _p(a), _p(t))

    def reset(self):
        obs = np.empty((self.E, self.D), dtype=np.float32)
        lib().np_oracle_reset(self._h, _p(obs))
        return obs

    def vec_step(self, actions, auto_reset=True):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.E, self.N)
        obs = np.empty((self.E, self.D), dtype=np.float32)
        term = np.zeros((self.E, self.D), dtype=np.float32)
        rew = np.empty(self.E)
        done = np.empty(self.E, dtype=np.uint8)
        lib().np_oracle_vec_step(self._h, _p(a), _p(obs), _p(rew), _p(done), _p(term),
                                 C.c_int(int(auto_reset)))
        return obs, rew, done.astype(bool), term

    def step(self, actions):
        obs, rew, done, _ = self.vec_step(actions, auto_reset=False)
        return obs, rew, done

    def state(self):
        E, N = self.E, self.N
        s = dict(amount=np.empty(E), amount_tag=np.empty(E, np.int32), total_asset=np.empty(E),
                 ta_tag=np.empty(E, np.int32), gamma_reward=np.empty(E),
                 g_tag=np.empty(E, np.int32), episode_return=np.empty(E),
                 day=np.empty(E, np.int32), stocks=np.empty((E, N), np.float32),
                 cool_down=np.empty((E, N), np.float32))
        lib().np_oracle_get_state(self._h, _p(s["amount"]), _p(s["amount_tag"]),
                                  _p(s["total_asset"]), _p(s["ta_tag"]), _p(s["gamma_reward"]),
                                  _p(s["g_tag"]), _p(s["episode_return"]), _p(s["day"]),
                                  _p(s["stocks"]), _p(s["cool_down"]))
        return s
