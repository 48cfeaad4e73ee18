"""ctypes wrapper over oracle/stoploss_oracle.c -- TEST INFRASTRUCTURE (see stock.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .stock import lib, _p


class SlCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_assets", C.c_int32), ("n_cols", C.c_int32),
                ("n_days", C.c_int32), ("discrete_actions", C.c_int32),
                ("shares_increment", C.c_int32), ("use_turbulence", C.c_int32),
                ("patient", C.c_int32), ("hmax", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("initial_amount", C.c_double),
                ("cash_penalty_proportion", C.c_double), ("turbulence_threshold", C.c_double),
                ("stoploss_penalty", C.c_double), ("min_profit_penalty", C.c_double)]


SCALARS = ("coh", "turbulence", "sum_trades", "logged_total", "logged_cash", "actual_num_trades")
VECTORS = ("holdings", "prev_holdings", "closing_diff_avg_buy", "profit_sell_diff_avg_buy",
           "n_buys", "avg_buy_price")


class StopLossOracle:
    """close [T,N], info [T,N,C] (daily_information_cols per asset, ticker-major), turb [T]."""

    def __init__(self, close, info, turb=None, *, n_envs=1, buy_cost_pct=3e-3, sell_cost_pct=3e-3,
                 hmax=10, discrete_actions=False, shares_increment=1, stoploss_penalty=0.9,
                 profit_loss_ratio=2, turbulence_threshold=None, initial_amount=1e6,
                 cash_penalty_proportion=0.1, patient=False):
        self.close = np.ascontiguousarray(close, dtype=np.float64)
        T, N = self.close.shape
        self.info = np.ascontiguousarray(info, dtype=np.float64).reshape(T, N, -1)
        self.turb = np.ascontiguousarray(np.zeros(T) if turb is None else turb, dtype=np.float64)
        self.E, self.N, self.Cc, self.T = int(n_envs), N, self.info.shape[2], T
        self.D = 1 + N + N * self.Cc
        L = lib()
        L.sl_oracle_create.restype = C.c_void_p
        min_profit_penalty = 1 + profit_loss_ratio * (1 - stoploss_penalty)      # :101
        self.cfg = SlCfg(self.E, N, self.Cc, T, int(discrete_actions), int(shares_increment),
                         int(turbulence_threshold is not None), int(patient), float(hmax),
                         float(buy_cost_pct), float(sell_cost_pct), float(initial_amount),
                         float(cash_penalty_proportion),
                         float(turbulence_threshold if turbulence_threshold is not None else 0),
                         float(stoploss_penalty), float(min_profit_penalty))
        self._h = C.c_void_p(L.sl_oracle_create(C.byref(self.cfg), _p(self.close), _p(self.info),
                                                _p(self.turb)))

    def __del__(self):
        try:
            if self._h:
                lib().sl_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _starts(self, starts):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(
            0 if starts is None else starts, np.int32), (self.E,)))

    def reset(self, starts=None):
        obs = np.empty((self.E, self.D))
        lib().sl_oracle_reset(self._h, _p(self._starts(starts)), _p(obs))
        return obs

    def vec_step(self, actions, starts=None, auto_reset=True):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.E, self.N)
        obs = np.empty((self.E, self.D))
        term = np.zeros((self.E, self.D))
        rew = np.empty(self.E)
        done = np.empty(self.E, dtype=np.uint8)
        lib().sl_oracle_vec_step(self._h, _p(a), _p(obs), _p(rew), _p(done), _p(term),
                                 _p(self._starts(starts)), C.c_int(int(auto_reset)))
        return obs, rew, done.astype(bool), term

    def step(self, actions):
        obs, rew, done, _ = self.vec_step(actions, auto_reset=False)
        return obs, rew, done

    def state(self):
        E, N = self.E, self.N
        scal, vec, ints = np.empty((6, E)), np.empty((6, E, N)), np.empty((3, E), np.int32)
        lib().sl_oracle_get_state(self._h, _p(scal), _p(vec), _p(ints))
        s = {k: scal[j] for j, k in enumerate(SCALARS)}
        s.update({k: vec[j] for j, k in enumerate(VECTORS)})
        s.update(date_index=ints[0], start=ints[1], episode=ints[2])
        return s
