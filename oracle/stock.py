"""ctypes wrapper over oracle/stock_oracle.c -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  See the header of stock_oracle.c for the parity status and the
reference lines each function follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.stock_oracle_create.restype = C.c_void_p
        _LIB.stock_oracle_obs_dim.restype = C.c_int
    return _LIB


class StockCfg(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_tech", C.c_int32),
        ("n_days", C.c_int32), ("hmax", C.c_int32), ("use_turbulence", C.c_int32),
        ("reset_quirk", C.c_int32), ("initial", C.c_int32),
        ("single_ticker", C.c_int32), ("reserved0", C.c_int32),
        ("buy_cost_pct", C.c_double), ("sell_cost_pct", C.c_double),
        ("reward_scaling", C.c_double), ("turbulence_threshold", C.c_double),
    ]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class StockOracle:
    """Batch of E scalar envs following env_stocktrading.py:24-485 (reference tree).

    close [T,N] f64, tech [T,K,N] f64, risk [T] f64.  cash0 scalar or [E];
    shares0 [N] or [E,N].
    """

    def __init__(self, close, tech, risk, *, n_envs=1, hmax=100, initial_amount=1_000_000,
                 num_stock_shares=None, buy_cost_pct=1e-3, sell_cost_pct=1e-3,
                 reward_scaling=1e-4, turbulence_threshold=None, reset_quirk=True,
                 initial=True, day=0):
        self.close = np.ascontiguousarray(close, dtype=np.float64)
        T, N = self.close.shape
        tech = np.asarray(tech, dtype=np.float64).reshape(T, -1, N) if np.size(tech) else \
            np.zeros((T, 0, N))
        self.tech = np.ascontiguousarray(tech)
        K = self.tech.shape[1]
        self.risk = np.ascontiguousarray(
            np.zeros(T) if risk is None else risk, dtype=np.float64)
        self.E, self.N, self.K, self.T = int(n_envs), N, K, T
        self.D = 1 + 2 * N + K * N
        self.cfg = StockCfg(self.E, N, K, T, int(hmax), int(turbulence_threshold is not None),
                            int(bool(reset_quirk)), int(bool(initial)),
                            # one ticker in the frame: the reference's single-stock branches
                            # (env_stocktrading.py:415-422, :443-450, :470-476)
                            int(N == 1), 0, float(buy_cost_pct),
                            float(sell_cost_pct), float(reward_scaling),
                            float(turbulence_threshold if turbulence_threshold is not None else 0.0))
        L = lib()
        self._h = C.c_void_p(L.stock_oracle_create(C.byref(self.cfg), _p(self.close),
                                                   _p(self.tech), _p(self.risk)))
        cash0 = np.broadcast_to(np.asarray(initial_amount, dtype=np.float64), (self.E,)).copy()
        if num_stock_shares is None:
            num_stock_shares = np.zeros(N, dtype=np.int64)
        sh0 = np.ascontiguousarray(
            np.broadcast_to(np.asarray(num_stock_shares, dtype=np.int64), (self.E, N)))
        L.stock_oracle_init(self._h, _p(cash0), _p(sh0), C.c_int(int(day)))

    def __del__(self):
        try:
            if self._h:
                lib().stock_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        obs = np.empty((self.E, self.D), dtype=np.float64)
        lib().stock_oracle_reset(self._h, _p(obs))
        return obs

    def _act(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.E, self.N)
        return a

    def step(self, actions, want_obs=True, want_realised=False):
        """gym semantics (no auto-reset) -> obs f64 [E,D], reward f64 [E], done bool [E]."""
        a = self._act(actions)
        obs = np.empty((self.E, self.D), dtype=np.float64) if want_obs else None
        rew = np.empty(self.E, dtype=np.float64)
        done = np.empty(self.E, dtype=np.uint8)
        real = np.empty((self.E, self.N), dtype=np.int64) if want_realised else None
        lib().stock_oracle_step(self._h, _p(a), _p(obs) if want_obs else None, _p(rew), _p(done),
                                _p(real) if want_realised else None)
        out = (obs, rew, done.astype(bool))
        return out + (real,) if want_realised else out

    def vec_step(self, actions, want_obs=True):
        """SB3 DummyVecEnv semantics -> obs, reward, done, terminal_obs (rows valid where done)."""
        a = self._act(actions)
        obs = np.empty((self.E, self.D), dtype=np.float64) if want_obs else None
        term = np.zeros((self.E, self.D), dtype=np.float64) if want_obs else None
        rew = np.empty(self.E, dtype=np.float64)
        done = np.empty(self.E, dtype=np.uint8)
        lib().stock_oracle_vec_step(self._h, _p(a), _p(obs) if want_obs else None, _p(rew),
                                    _p(done), _p(term) if want_obs else None)
        return obs, rew, done.astype(bool), term

    def state(self):
        E, N = self.E, self.N
        s = dict(cash=np.empty(E), shares=np.empty((E, N), dtype=np.int64),
                 day=np.empty(E, dtype=np.int32), price_day=np.empty(E, dtype=np.int32),
                 cost=np.empty(E), trades=np.empty(E, dtype=np.int32),
                 last_reward=np.empty(E), turbulence=np.empty(E),
                 episode=np.empty(E, dtype=np.int32))
        lib().stock_oracle_get_state(self._h, _p(s["cash"]), _p(s["shares"]), _p(s["day"]),
                                     _p(s["price_day"]), _p(s["cost"]), _p(s["trades"]),
                                     _p(s["last_reward"]), _p(s["turbulence"]), _p(s["episode"]))
        return s

    def episode_stats(self):
        out = np.empty((self.E, 6), dtype=np.float64)
        lib().stock_oracle_episode_stats(self._h, _p(out))
        return out
