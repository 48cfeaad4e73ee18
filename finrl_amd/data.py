"""The data formats on either side of the env path: the DataFrame index
contract the DataFrame-backed envs rely on and the array layout the array-state env consumes.
Vectorised restatements -- same results as the reference helpers, no network, no stockstats.

  data_split   <- finrl/meta/preprocessor/preprocessors.py:24-33
  df_to_array  <- finrl/meta/data_processors/processor_yahoofinance.py:293-318 plus the
                  NaN / inf -> 0 clean-up of finrl/meta/data_processor.py:74-84
  clean_data   <- finrl/meta/preprocessor/preprocessors.py:107-131 (FeatureEngineer.clean_data:
                  keep only tickers with a close on every date -- what makes the panel complete
                  before add_turbulence / the envs see it)
"""
from __future__ import annotations

import numpy as np


def data_split(df, start, end, target_date_col="date"):
    """Rows with start <= date < end, sorted by (date, tic), index = day ordinal."""
    data = df[(df[target_date_col] >= start) & (df[target_date_col] < end)]
    data = data.sort_values([target_date_col, "tic"], ignore_index=True)
    data.index = data[target_date_col].factorize()[0]
    return data


def clean_data(data):
    """Rows sorted by (date, tic), index = day ordinal, tickers with a missing close on any
    date dropped (delisted / not yet listed)."""
    df = data.copy()
    df = df.sort_values(["date", "tic"], ignore_index=True)
    df.index = df.date.factorize()[0]
    closes = df.pivot_table(index="date", columns="tic", values="close")
    keep = closes.columns[closes.notna().all(axis=0)]
    return df[df.tic.isin(keep)]


def df_to_array(df, tech_indicator_list, if_vix, price_col="adjcp"):
    """-> price_array [T, N], tech_array [T, N*K] (ticker-major: all indicators of ticker 0,
    then ticker 1, ...), turbulence_array [T] -- the config of env_stocktrading_np.

    The reference hstacks one ticker at a time in `df.tic.unique()` order, each ticker's rows
    in frame order; this does the same with one stable sort."""
    tics = df["tic"].to_numpy()
    uniq, first = np.unique(tics, return_index=True)
    uniq = uniq[np.argsort(first)]                      # order of first appearance
    code = {t: i for i, t in enumerate(uniq)}
    tic_code = np.fromiter((code[t] for t in tics), dtype=np.int64, count=len(tics))
    order = np.argsort(tic_code, kind="stable")         # ticker-major, frame order within
    N = len(uniq)
    if len(df) % N:
        raise ValueError("every ticker must have the same number of rows")
    T = len(df) // N
    price = df[price_col].to_numpy(dtype=np.float64)[order].reshape(N, T).T
    tech = df[list(tech_indicator_list)].to_numpy(dtype=np.float64)[order]
    K = len(tech_indicator_list)
    tech = tech.reshape(N, T, K).transpose(1, 0, 2).reshape(T, N * K)
    risk_col = "vix" if if_vix else "turbulence"
    turb = df[risk_col].to_numpy()[order][:T]           # the reference takes the FIRST ticker's
    tech = np.array(tech)                               # rows (:304-307)
    tech[np.isnan(tech)] = 0
    tech[np.isinf(tech)] = 0
    return np.ascontiguousarray(price), tech, np.array(turb)
