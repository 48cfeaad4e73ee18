"""HIP risk precompute (through the C ABI) vs the reference fixtures and the NumPy oracle.

Floating point, tolerances stated here: covariance |d| <= 1e-12 x max|cov| (different summation
order than BLAS); turbulence index rel 1e-8 (Jacobi eigen-decomposition on the GPU vs LAPACK SVD
inside np.linalg.pinv; measured 2e-15 on the fixtures); zero / non-zero pattern (the first-two-positive rule,
preprocessors.py:247-257) exact."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("riskpre_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "riskpre_*.npz")))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.mark.parametrize("name", NAMES)
def test_turbulence_matches_reference_fixture(name):
    _need_gpu()
    from finrl_amd import riskpre
    z = np.load(os.path.join(GOLDEN, f"riskpre_{name}.npz"), allow_pickle=False)
    got = riskpre.calculate_turbulence(z["close"]).cpu().numpy()
    ref = z["turbulence"]
    np.testing.assert_array_equal(got == 0, ref == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-8)
    print(name, "max rel", np.max(np.abs(got - ref)[ref > 0] / ref[ref > 0]))


@pytest.mark.parametrize("name", NAMES)
def test_rolling_cov_matches_reference_fixture(name):
    _need_gpu()
    from finrl_amd import riskpre
    z = np.load(os.path.join(GOLDEN, f"riskpre_{name}.npz"), allow_pickle=False)
    cov = riskpre.rolling_covariance(z["close"], int(z["lookback"])).cpu().numpy()
    assert cov.shape[0] == z["close"].shape[0] - int(z["lookback"])
    scale = np.abs(z["cov"]).max()
    np.testing.assert_allclose(cov[z["cov_index"]], z["cov"], rtol=0, atol=1e-12 * scale)
    np.testing.assert_array_equal(cov, cov.transpose(0, 2, 1))


def test_turbulence_matches_oracle_short_window_and_singular():
    """window 40 on 12 assets, two of them identical (singular covariance: the pseudo-inverse
    drops the null direction, as np.linalg.pinv does)."""
    _need_gpu()
    from finrl_amd import riskpre
    from oracle import riskpre as orc
    rng = np.random.default_rng(5)
    T, N = 120, 12
    close = 30 * np.exp(np.cumsum(rng.normal(0, 0.02, (T, N)), axis=0))
    close[:, 7] = close[:, 3] * 2.0
    got, quad = riskpre.calculate_turbulence(close, window=40, return_quadratic_forms=True)
    ref_q = orc.turbulence_quadratic_forms(close, 40)
    np.testing.assert_allclose(quad.cpu().numpy()[40:], ref_q[40:], rtol=1e-6)
    ref = orc.suppress_first_two(ref_q, 40)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-6)
    cov = riskpre.rolling_covariance(close, 40).cpu().numpy()
    np.testing.assert_allclose(cov, orc.rolling_covariance(close, 40), rtol=0,
                               atol=1e-12 * np.abs(cov).max())


def test_add_turbulence_and_cov_list_frames():
    _need_gpu()
    import pandas as pd
    from finrl_amd import riskpre
    z = np.load(os.path.join(GOLDEN, "riskpre_small.npz"), allow_pickle=False)
    T, N = z["close"].shape
    dates = pd.bdate_range("2015-01-01", periods=T).strftime("%Y-%m-%d")
    df = pd.DataFrame({"date": np.repeat(dates, N), "tic": np.tile([f"T{i}" for i in range(N)], T),
                       "close": z["close"].reshape(-1)})
    out = riskpre.add_turbulence(df)
    assert len(out) == len(df)
    np.testing.assert_allclose(out["turbulence"].to_numpy()[::N], z["turbulence"], rtol=1e-8)
    oc = riskpre.add_cov_list(df)
    assert len(oc) == (T - 252) * N and oc["cov_list"].iloc[0].shape == (N, N)
    with pytest.raises(Exception):
        riskpre.add_turbulence(df.iloc[:-1])           # ragged panel: fail loudly
    with pytest.raises(ValueError):
        riskpre.calculate_turbulence(z["close"][:100])
