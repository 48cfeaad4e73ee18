"""ctypes wrapper over oracle/crypto_oracle.c -- TEST INFRASTRUCTURE (see stock.py)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .stock import lib, _p


class CrCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_assets", C.c_int32), ("n_tech", C.c_int32),
                ("n_steps", C.c_int32), ("lookback", C.c_int32), ("reserved", C.c_int32),
                ("initial_cash", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("gamma", C.c_double)]


def action_norm_vector(price0):
    """env_multiple_crypto.py:103-111 evaluated with Python's own math (math.log(p, 10))."""
    out = []
    for price in np.asarray(price0, dtype=np.float64):
        x = math.floor(math.log(price, 10))
        out.append(1 / ((10) ** x))
    return np.asarray(out) * 10000


class CryptoOracle:
    def __init__(self, price, tech, *, n_envs=1, lookback=1, initial_capital=1e6,
                 buy_cost_pct=1e-3, sell_cost_pct=1e-3, gamma=0.99):
        self.price = np.ascontiguousarray(price, dtype=np.float64)
        self.tech = np.ascontiguousarray(tech, dtype=np.float64)
        T, N = self.price.shape
        self.E, self.N, self.W, self.T, self.L = int(n_envs), N, self.tech.shape[1], T, lookback
        self.D = 1 + N + self.W * lookback
        self.max_step = T - lookback - 1
        L = lib()
        L.cr_oracle_create.restype = C.c_void_p
        self.cfg = CrCfg(self.E, N, self.W, T, lookback, 0, float(initial_capital),
                         float(buy_cost_pct), float(sell_cost_pct), float(gamma))
        self._h = C.c_void_p(L.cr_oracle_create(C.byref(self.cfg), _p(self.price), _p(self.tech)))
        self.norm = action_norm_vector(self.price[0])
        L.cr_oracle_set_norm(self._h, _p(np.ascontiguousarray(self.norm)))

    def __del__(self):
        try:
            if self._h:
                lib().cr_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        obs = np.empty((self.E, self.D), dtype=np.float32)
        lib().cr_oracle_reset(self._h, _p(obs))
        return obs

    def reset_masked(self, mask):
        """reset() (:48-57) for the envs with mask[e] != 0 only -> their observation rows
        (rows of the other envs are left zero)."""
        obs = np.zeros((self.E, self.D), dtype=np.float32)
        L = lib()
        for e in np.flatnonzero(np.asarray(mask)):
            row = np.empty(self.D, dtype=np.float32)
            L.cr_oracle_reset_env(self._h, C.c_int(int(e)), _p(row))
            obs[e] = row
        return obs

    def vec_step(self, actions, auto_reset=True):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.E, self.N)
        obs = np.empty((self.E, self.D), dtype=np.float32)
        term = np.zeros((self.E, self.D), dtype=np.float32)
        rew = np.empty(self.E)
        done = np.empty(self.E, dtype=np.uint8)
        lib().cr_oracle_vec_step(self._h, _p(a), _p(obs), _p(rew), _p(done), _p(term),
                                 C.c_int(int(auto_reset)))
        return obs, rew, done.astype(bool), term

    def step(self, actions):
        obs, rew, done, _ = self.vec_step(actions, auto_reset=False)
        return obs, rew, done

    def state(self):
        E, N = self.E, self.N
        s = dict(cash=np.empty(E), total_asset=np.empty(E), gamma_return=np.empty(E),
                 episode_return=np.empty(E), time=np.empty(E, dtype=np.int32),
                 stocks=np.empty((E, N), dtype=np.float32))
        lib().cr_oracle_get_state(self._h, _p(s["cash"]), _p(s["total_asset"]),
                                  _p(s["gamma_return"]), _p(s["episode_return"]), _p(s["time"]),
                                  _p(s["stocks"]))
        return s
