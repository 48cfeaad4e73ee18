"""data_split / df_to_array restatements vs the reference functions (build container only:
skipped where /root/reference is absent) and vs their documented contracts everywhere."""
import os

import numpy as np
import pandas as pd
import pytest

from finrl_amd.data import data_split, df_to_array


def _frame(T=9, N=4, K=3, seed=0):
    rng = np.random.default_rng(seed)
    dates = pd.date_range("2020-01-01", periods=T).strftime("%Y-%m-%d")
    tics = [f"T{i}" for i in (2, 0, 3, 1)][:N]
    rows = []
    for d in dates:
        for t in tics:
            r = {"date": d, "time": d, "tic": t, "close": rng.uniform(10, 200),
                 "adjcp": rng.uniform(10, 200), "turbulence": rng.uniform(0, 100),
                 "vix": rng.uniform(10, 40)}
            for k in range(K):
                r[f"ind{k}"] = rng.normal()
            rows.append(r)
    df = pd.DataFrame(rows)
    df.loc[3, "ind1"] = np.nan
    df.loc[5, "ind2"] = np.inf
    return df.sample(frac=1.0, random_state=1).reset_index(drop=True)


def test_data_split_contract():
    df = _frame()
    out = data_split(df, "2020-01-03", "2020-01-07")
    assert out.index[0] == 0 and out.index.max() == 3
    assert list(out["date"].unique()) == ["2020-01-03", "2020-01-04", "2020-01-05", "2020-01-06"]
    for day in range(4):
        assert list(out.loc[day, "tic"]) == sorted(out.loc[day, "tic"])


def test_df_to_array_contract():
    df = _frame().sort_values(["time", "tic"]).reset_index(drop=True)
    names = ["ind0", "ind1", "ind2"]
    price, tech, turb = df_to_array(df, names, if_vix=False)
    T, N, K = 9, 4, 3
    assert price.shape == (T, N) and tech.shape == (T, N * K) and turb.shape == (T,)
    uniq = list(df.tic.unique())
    for j, t in enumerate(uniq):
        sub = df[df.tic == t]
        np.testing.assert_array_equal(price[:, j], sub["adjcp"].values)
        ref = sub[names].values.copy()
        ref[~np.isfinite(ref)] = 0
        np.testing.assert_array_equal(tech[:, j * K:(j + 1) * K], ref)
    np.testing.assert_array_equal(turb, df[df.tic == uniq[0]]["turbulence"].values)
    assert np.isfinite(tech).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference/finrl"), reason="reference not present")
def test_matches_reference_functions():
    """Against the reference's own data_split (preprocessors.py:24-33; imported behind a
    stockstats stub) and the body of df_to_array as YahooFinanceProcessor defines it."""
    import sys
    import types
    from oracle import ref_harness as rh
    rh.install()
    sys.modules.setdefault("stockstats", types.SimpleNamespace(StockDataFrame=object))
    cfg = types.ModuleType("finrl.config")
    cfg.INDICATORS = []
    sys.modules.setdefault("finrl.config", cfg)
    yd = types.ModuleType("finrl.meta.preprocessor.yahoodownloader")
    yd.YahooDownloader = object
    sys.modules.setdefault("finrl.meta.preprocessor.yahoodownloader", yd)
    import importlib
    sys.modules["finrl"].config = cfg
    pre = importlib.import_module("finrl.meta.preprocessor.preprocessors")
    df = _frame()
    a = pre.data_split(df, "2020-01-02", "2020-01-08")
    b = data_split(df, "2020-01-02", "2020-01-08")
    pd.testing.assert_frame_equal(a, b)
    # FeatureEngineer.clean_data (:107-131) on a ragged frame: one ticker misses a date, one a close
    from finrl_amd.data import clean_data
    t_a, t_b = sorted(df.tic.unique())[:2]
    ragged = df.drop(df.index[(df.tic == t_a) & (df.date == df.date.max())]).copy()
    ragged.loc[(ragged.tic == t_b) & (ragged.date == ragged.date.min()), "close"] = np.nan
    fe = pre.FeatureEngineer(use_technical_indicator=False)
    pd.testing.assert_frame_equal(fe.clean_data(ragged), clean_data(ragged))
    assert clean_data(ragged).tic.nunique() == df.tic.nunique() - 2


def test_panel_npz_round_trip(tmp_path):
    """the .npz panel format -- plain arrays (no pickle), every field back bit for
    bit, the packed device images of the reloaded panel equal the original's."""
    from finrl_amd.panel import StockPanel
    df = data_split(_frame(T=7, N=4, K=3).fillna(0).replace(np.inf, 0), "2020-01-01", "2020-01-08")
    a = StockPanel.from_dataframe(df, ["ind0", "ind1", "ind2"], "turbulence")
    path = os.path.join(tmp_path, "panel.npz")
    a.save(path)
    with np.load(path, allow_pickle=False) as z:           # plain arrays only
        assert str(z["format"]) == StockPanel.NPZ_FORMAT and z["close"].dtype == np.float64
    b = StockPanel.load(path)
    for k in ("close", "tech", "risk"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    assert (a.tech_names, a.dates, a.tickers) == (b.tech_names, b.dates, b.tickers)
    np.testing.assert_array_equal(a.obs_template(), b.obs_template())
    np.testing.assert_array_equal(a.signed_close(), b.signed_close())
    np.savez(os.path.join(tmp_path, "other.npz"), format=np.array("something else"), close=a.close)
    with pytest.raises(ValueError):
        StockPanel.load(os.path.join(tmp_path, "other.npz"))
