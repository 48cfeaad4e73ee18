"""Risk precompute on the GPU, through the C ABI (finenv_riskpre_*):

* calculate_turbulence / add_turbulence  <- FeatureEngineer, finrl/meta/preprocessor/
  preprocessors.py:203-267
* rolling_covariance / add_cov_list      <- tutorials/2-Advance/
  FinRL_PortfolioAllocation_Explainable_DRL.py:160-172

Contract: a complete panel (every ticker on every date, no NaN) -- a ragged frame raises instead
of silently taking another path (there is no CPU path in this package)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat


def _pivot_close(df):
    piv = df.pivot(index="date", columns="tic", values="close")       # :218
    if piv.isna().any().any():
        raise nat.FinenvError("risk precompute needs a complete panel (every ticker on every "
                              "date); drop or fill the missing rows first")
    return piv


def _dev_close(close, device):
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
    if isinstance(close, torch.Tensor):
        t = close.to(device=dev, dtype=torch.float64).contiguous()
    else:
        a = np.ascontiguousarray(close, dtype=np.float64)
        if not np.isfinite(a).all():
            raise nat.FinenvError("close contains NaN/inf")
        t = torch.from_numpy(a).to(dev)
    if t.dim() != 2:
        raise nat.FinenvError("close must be [T, N]")
    return t


def _returns(close_t):
    import torch
    T, N = close_t.shape
    ret = torch.empty_like(close_t)
    stream = C.c_void_p(torch.cuda.current_stream(close_t.device).cuda_stream)
    nat.check(nat.lib().finenv_riskpre_returns(C.c_void_p(close_t.data_ptr()),
                                               C.c_void_p(ret.data_ptr()), T, N, stream),
              None, "finenv_riskpre_returns")
    return ret, stream


def calculate_turbulence(close, window=252, device="cuda", return_quadratic_forms=False):
    """close [T, N] (array or tensor) -> turbulence index [T] float64 CUDA tensor."""
    import torch
    close_t = _dev_close(close, device)
    T, N = close_t.shape
    if T < window:
        raise ValueError("Turbulence information could not be added.")            # :265-266
    ret, stream = _returns(close_t)
    quad = torch.zeros(T, dtype=torch.float64, device=close_t.device)
    out = torch.empty(T, dtype=torch.float64, device=close_t.device)
    nat.check(nat.lib().finenv_riskpre_turbulence(
        C.c_void_p(ret.data_ptr()), C.c_void_p(quad.data_ptr()), C.c_void_p(out.data_ptr()),
        T, N, int(window), stream), None, "finenv_riskpre_turbulence")
    return (out, quad) if return_quadratic_forms else out


def rolling_covariance(close, lookback=252, device="cuda"):
    """close [T, N] -> cov_list [T - lookback, N, N] float64 CUDA tensor."""
    import torch
    close_t = _dev_close(close, device)
    T, N = close_t.shape
    if T <= lookback:
        raise ValueError(f"need more than lookback={lookback} days, got {T}")
    ret, stream = _returns(close_t)
    cov = torch.empty(T - lookback, N, N, dtype=torch.float64, device=close_t.device)
    nat.check(nat.lib().finenv_riskpre_rolling_cov(
        C.c_void_p(ret.data_ptr()), C.c_void_p(cov.data_ptr()), T, N, int(lookback), stream),
        None, "finenv_riskpre_rolling_cov")
    return cov


def add_turbulence(df, window=252, device="cuda"):
    """FeatureEngineer.add_turbulence (:203-213): merge a `turbulence` column on date."""
    import pandas as pd
    piv = _pivot_close(df)
    turb = calculate_turbulence(piv.to_numpy(np.float64), window, device).cpu().numpy()
    idx = pd.DataFrame({"date": piv.index, "turbulence": turb})
    out = df.copy().merge(idx, on="date")
    return out.sort_values(["date", "tic"]).reset_index(drop=True)


def add_cov_list(df, lookback=252, device="cuda"):
    """Tutorial :157-176: one [N, N] covariance per date from `lookback` on, merged on date
    (dates before `lookback` are dropped by the inner merge, as in the tutorial)."""
    import pandas as pd
    d = df.sort_values(["date", "tic"], ignore_index=True)
    piv = _pivot_close(d)
    cov = rolling_covariance(piv.to_numpy(np.float64), lookback, device).cpu().numpy()
    df_cov = pd.DataFrame({"date": piv.index[lookback:], "cov_list": list(cov)})
    return d.merge(df_cov, on="date").sort_values(["date", "tic"]).reset_index(drop=True)
