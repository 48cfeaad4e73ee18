"""oracle/riskpre.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's risk precompute steps:

* calculate_turbulence  <- finrl/meta/preprocessor/preprocessors.py:215-267
  (FeatureEngineer.calculate_turbulence: 252-day rolling covariance of daily returns, Moore-
  Penrose inverse, quadratic form of the current day's de-meaned return; first two positive
  values suppressed, :250-257)
* rolling_covariance    <- tutorials/2-Advance/FinRL_PortfolioAllocation_Explainable_DRL.py:160-172
  (`cov_list`: covariance of the `lookback` returns ending at day i, inclusive)

Parity status: PINNED -- tests/golden/riskpre_*.npz hold outputs of the unmodified
FeatureEngineer.calculate_turbulence and of the tutorial's five pandas lines run in the build
container.  Contract of the restatement: a complete panel (every ticker on every date, no NaN in
close), so the reference's missing-ticker filtering (:234-237) reduces to dropping the leading NaN
return row.  Floating point: covariance via np.cov and the inverse via np.linalg.pinv, i.e. the
same library calls pandas makes for a NaN-free frame.
"""
from __future__ import annotations

import numpy as np


def pct_change(close):
    """DataFrame.pct_change(): p[t] / p[t-1] - 1, first row NaN (:221)."""
    close = np.asarray(close, dtype=np.float64)
    r = np.full_like(close, np.nan)
    r[1:] = close[1:] / close[:-1] - 1
    return r


def turbulence_quadratic_forms(close, window=252):
    """temp of :244-246 for every day i >= window (0 before)."""
    r = pct_change(close)
    T = r.shape[0]
    out = np.zeros(T)
    for i in range(window, T):
        hist = r[i - window:i]                                   # :229-232
        lead = int(np.isnan(hist).sum(axis=0).min())             # :234-236
        hist = hist[lead:]
        cov = np.cov(hist.T, ddof=1)                             # pandas .cov() fast path, :238
        cur = r[i] - np.mean(hist, axis=0)                       # :239-241
        out[i] = cur.dot(np.linalg.pinv(cov)).dot(cur)           # :244-246
    return out


def suppress_first_two(quad, window=252):
    """:247-257 -- the first two positive values are reported as 0."""
    out = np.zeros_like(quad)
    count = 0
    for i in range(window, len(quad)):
        if quad[i] > 0:
            count += 1
            if count > 2:
                out[i] = quad[i]
    return out


def calculate_turbulence(close, window=252):
    close = np.asarray(close, dtype=np.float64)
    if close.shape[0] < window:
        raise ValueError("Turbulence information could not be added.")    # :265-266
    return suppress_first_two(turbulence_quadratic_forms(close, window), window)


def rolling_covariance(close, lookback=252):
    """cov_list[i - lookback] for i in range(lookback, T): returns of price rows [i-lookback, i]
    (both inclusive, `df.loc[i-lookback:i]`), tutorial :162-168."""
    r = pct_change(close)
    T, N = r.shape
    out = np.empty((max(T - lookback, 0), N, N))
    for i in range(lookback, T):
        out[i - lookback] = np.cov(r[i - lookback + 1:i + 1].T, ddof=1)
    return out
