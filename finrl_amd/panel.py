"""Panel packer: the DataFrame / arrays the reference env consumes -> the device-resident
structure-of-arrays the HIP kernels read.

Reference contracts followed:
  * DataFrame form (env_stocktrading.py:64, :336 `df.loc[day]`): integer index = day
    ordinal (finrl/meta/preprocessor/preprocessors.py:24-33 `data_split`), N rows per day in
    ticker order, columns close / <tech names> / <risk col>.
  * observation order (env_stocktrading.py:456-467): [cash | close[N] | shares[N] |
    tech_0[N] | ... | tech_{K-1}[N]]  (indicator-major).

Device layout (see include/finenv.h `finenv_stock_panel`):
  close      f64 [T][N]   money arithmetic runs on the reference's own doubles; the SIGN BIT carries
                          the day's "untradable" flag: set <=> tech_0[t][i] == 1.0 evaluated in
                          fp64 (:105, :174)
  obs_tmpl   f32 [T][D]   ready-made observation rows (cash / holdings slots zero)
  risk       f64 [T]
The whole DOW30 x 8 x 2893-day panel is 4.2 MB: it stays resident in L2 / Infinity Cache,
so per-step HBM traffic is the per-env state, actions and observations only.
"""
from __future__ import annotations

import numpy as np


class StockPanel:
    def __init__(self, close, tech=None, risk=None, *, tech_names=None, dates=None,
                 tickers=None):
        close = np.ascontiguousarray(close, dtype=np.float64)
        if close.ndim != 2:
            raise ValueError("close must be [T, N]")
        T, N = close.shape
        if tech is None or np.size(tech) == 0:
            tech = np.zeros((T, 0, N), dtype=np.float64)
        tech = np.ascontiguousarray(tech, dtype=np.float64)
        if tech.shape[0] != T or tech.shape[-1] != N or tech.ndim != 3:
            raise ValueError(f"tech must be [T, K, N] = [{T}, K, {N}], got {tech.shape}")
        risk = np.zeros(T) if risk is None else np.ascontiguousarray(risk, dtype=np.float64)
        if risk.shape != (T,):
            raise ValueError("risk must be [T]")
        self.close, self.tech, self.risk = close, tech, risk
        self.T, self.N, self.K = T, N, tech.shape[1]
        self.D = 1 + 2 * N + self.K * N
        self.tech_names = list(tech_names) if tech_names is not None else \
            [f"tech{k}" for k in range(self.K)]
        self.dates = list(dates) if dates is not None else list(range(T))
        self.tickers = list(tickers) if tickers is not None else [f"TIC{i}" for i in range(N)]
        self._device_cache = {}

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_dataframe(cls, df, tech_indicator_list, risk_indicator_col="turbulence"):
        """Pack the frame a reference StockTradingEnv would be given."""
        idx = np.asarray(df.index)
        days, inv = np.unique(idx, return_inverse=True)
        T = len(days)
        if len(df) % T:
            raise ValueError("every day must hold the same number of tickers")
        N = len(df) // T
        order = np.argsort(inv, kind="stable")          # df.loc[day] keeps frame order
        if not np.array_equal(np.bincount(inv, minlength=T), np.full(T, N)):
            raise ValueError("every day must hold the same number of tickers")

        def col(name):
            return np.asarray(df[name], dtype=np.float64)[order].reshape(T, N)

        close = col("close")
        tech = np.stack([col(t) for t in tech_indicator_list], axis=1) if tech_indicator_list \
            else np.zeros((T, 0, N))
        if risk_indicator_col in df.columns:
            risk = col(risk_indicator_col)[:, 0]        # .values[0], :341
        else:
            risk = np.zeros(T)
        dates = np.asarray(df["date"])[order].reshape(T, N)[:, 0].tolist() \
            if "date" in df.columns else list(range(T))
        tickers = np.asarray(df["tic"])[order].reshape(T, N)[0].tolist() \
            if "tic" in df.columns else None
        return cls(close, tech, risk, tech_names=tech_indicator_list, dates=dates,
                   tickers=tickers)

    # ------------------------------------------------------------------ .npz panel format
    # the on-disk form of a packed panel.  Plain arrays only (loadable with
    # numpy.load's default allow_pickle=False): close [T,N] f64, tech [T,K,N] f64, risk [T] f64,
    # tech_names / dates / tickers as unicode arrays, format tag.
    NPZ_FORMAT = "finrl_amd.StockPanel/1"

    def save(self, path):
        np.savez_compressed(path, format=np.array(self.NPZ_FORMAT), close=self.close, tech=self.tech,
                            risk=self.risk, tech_names=np.asarray(self.tech_names, dtype=str),
                            dates=np.asarray([str(d) for d in self.dates], dtype=str),
                            tickers=np.asarray([str(t) for t in self.tickers], dtype=str))

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        if str(z["format"]) != cls.NPZ_FORMAT:
            raise ValueError(f"{path}: not a {cls.NPZ_FORMAT} file")
        return cls(z["close"], z["tech"], z["risk"], tech_names=z["tech_names"].tolist(),
                   dates=z["dates"].tolist(), tickers=z["tickers"].tolist())

    # ------------------------------------------------------------------ host packing
    def obs_template(self) -> np.ndarray:
        """f32 [T, D]: observation rows with cash / holdings slots left zero."""
        T, N, K = self.T, self.N, self.K
        out = np.zeros((T, self.D), dtype=np.float32)
        out[:, 1:1 + N] = self.close.astype(np.float32)
        out[:, 1 + 2 * N:] = self.tech.reshape(T, K * N).astype(np.float32)
        return out

    def signed_close(self) -> np.ndarray:
        """f64 [T, N]: the closes with the fork's "untradable" flag in the sign bit -- set iff the
        first indicator of that ticker equals 1.0 on that day (fp64 compare; :105, :174).  The
        kernels use |close| for every valuation and the sign only in the trade rules."""
        if self.N > 128:
            raise ValueError("the stock kernels support N <= 128 tickers")
        if (self.close < 0).any():
            raise ValueError("closes must be >= 0 (the sign bit is the untradable flag)")
        out = np.abs(self.close)              # also turns any -0.0 into +0.0
        if self.K:
            flag = self.tech[:, 0, :] == 1.0
            out = np.where(flag, np.copysign(out, -1.0), out)
        return np.ascontiguousarray(out)

    def to_device(self, device):
        """-> dict of torch tensors on `device` (cached per device)."""
        import torch
        key = str(device)
        if key not in self._device_cache:
            self._device_cache[key] = dict(
                close=torch.from_numpy(self.signed_close()).to(device),
                obs_tmpl=torch.from_numpy(self.obs_template()).to(device),
                risk=torch.from_numpy(self.risk).to(device),
            )
        return self._device_cache[key]

    def nbytes_device(self):
        return self.T * (8 * self.N + 4 * self.D + 8)


class PortfolioPanel:
    """Panel for StockPortfolioEnv (env_portfolio.py:105-112, :172-179): per day an N x N
    covariance matrix (`df["cov_list"]`, one object per row) and K indicator rows; the
    observation is their vertical stack, independent of per-env state.

    Device layout (include/finenv.h `finenv_portfolio_panel`):
      gross_ret f64 [T][N]   close[t+1]/close[t] - 1, evaluated elementwise in fp64 (:184)
      obs_tmpl  f32 [T][D]   D = (N + K) * N
    """

    def __init__(self, close, cov, tech=None, *, tech_names=None, dates=None, tickers=None):
        self.close = np.ascontiguousarray(close, dtype=np.float64)
        T, N = self.close.shape
        self.cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(T, N, N)
        if tech is None or np.size(tech) == 0:
            tech = np.zeros((T, 0, N))
        self.tech = np.ascontiguousarray(tech, dtype=np.float64).reshape(T, -1, N)
        self.T, self.N, self.K = T, N, self.tech.shape[1]
        self.D = (N + self.K) * N
        self.tech_names = list(tech_names) if tech_names is not None else \
            [f"tech{k}" for k in range(self.K)]
        self.dates = list(dates) if dates is not None else list(range(T))
        self.tickers = list(tickers) if tickers is not None else [f"TIC{i}" for i in range(N)]
        self._device_cache = {}

    @classmethod
    def from_dataframe(cls, df, tech_indicator_list):
        idx = np.asarray(df.index)
        days, inv = np.unique(idx, return_inverse=True)
        T = len(days)
        N = len(df) // T
        order = np.argsort(inv, kind="stable")
        col = lambda name: np.asarray(df[name], dtype=np.float64)[order].reshape(T, N)
        close = col("close")
        tech = np.stack([col(t) for t in tech_indicator_list], axis=1) if tech_indicator_list \
            else np.zeros((T, 0, N))
        covs = np.asarray(df["cov_list"], dtype=object)[order].reshape(T, N)[:, 0]
        cov = np.stack([np.asarray(c, dtype=np.float64) for c in covs])
        dates = np.asarray(df["date"])[order].reshape(T, N)[:, 0].tolist() \
            if "date" in df.columns else None
        tickers = np.asarray(df["tic"])[order].reshape(T, N)[0].tolist() \
            if "tic" in df.columns else None
        return cls(close, cov, tech, tech_names=tech_indicator_list, dates=dates, tickers=tickers)

    def obs_template(self):
        return np.concatenate([self.cov.reshape(self.T, -1), self.tech.reshape(self.T, -1)],
                              axis=1).astype(np.float32)

    def gross_returns(self):
        g = np.zeros((self.T, self.N))
        g[:-1] = (self.close[1:] / self.close[:-1]) - 1
        return g

    def to_device(self, device):
        import torch
        key = str(device)
        if key not in self._device_cache:
            self._device_cache[key] = dict(
                gross_ret=torch.from_numpy(self.gross_returns()).to(device),
                obs_tmpl=torch.from_numpy(self.obs_template()).to(device))
        return self._device_cache[key]
