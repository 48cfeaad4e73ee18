"""ctypes wrapper over oracle/portfolio_oracle.c -- TEST INFRASTRUCTURE (see stock.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .stock import lib, _p


class PfCfg(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_tech", C.c_int32),
                ("n_days", C.c_int32), ("initial_amount", C.c_double)]


class PortfolioOracle:
    """E scalar StockPortfolioEnv copies (env_portfolio.py:15-261).
    close [T,N], cov [T,N,N], tech [T,K,N] (float64)."""

    def __init__(self, close, cov, tech, *, n_envs=1, initial_amount=1_000_000):
        self.close = np.ascontiguousarray(close, dtype=np.float64)
        T, N = self.close.shape
        self.cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(T, N, N)
        self.tech = np.ascontiguousarray(tech, dtype=np.float64).reshape(T, -1, N)
        self.E, self.N, self.K, self.T = int(n_envs), N, self.tech.shape[1], T
        self.D = (N + self.K) * N
        L = lib()
        L.pf_oracle_create.restype = C.c_void_p
        self.cfg = PfCfg(self.E, N, self.K, T, float(initial_amount))
        self._h = C.c_void_p(L.pf_oracle_create(C.byref(self.cfg), _p(self.close), _p(self.cov),
                                                _p(self.tech)))

    def __del__(self):
        try:
            if self._h:
                lib().pf_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reset(self):
        obs = np.empty((self.E, self.D))
        lib().pf_oracle_reset(self._h, _p(obs))
        return obs

    def vec_step(self, actions, auto_reset=True, want_obs=True):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.E, self.N)
        obs = np.empty((self.E, self.D)) if want_obs else None
        term = np.zeros((self.E, self.D)) if want_obs else None
        rew = np.empty(self.E)
        done = np.empty(self.E, dtype=np.uint8)
        lib().pf_oracle_vec_step(self._h, _p(a), _p(obs) if want_obs else None, _p(rew), _p(done),
                                 _p(term) if want_obs else None, C.c_int(int(auto_reset)))
        return obs, rew, done.astype(bool), term

    def step(self, actions, want_obs=True):
        obs, rew, done, _ = self.vec_step(actions, auto_reset=False, want_obs=want_obs)
        return obs, rew, done

    def state(self):
        v = np.empty(self.E)
        d = np.empty(self.E, dtype=np.int32)
        lib().pf_oracle_get_state(self._h, _p(v), _p(d))
        return dict(value=v, day=d)
