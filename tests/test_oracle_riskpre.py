"""oracle/riskpre.py vs committed outputs of the unmodified reference
FeatureEngineer.calculate_turbulence (preprocessors.py:215-267) and the tutorial's cov_list lines
(tests/golden/riskpre_*.npz).  Float work: rtol 1e-9 on the turbulence index (the quadratic form
goes through an SVD-based pseudo-inverse), 1e-12 of the covariance scale on cov."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("riskpre_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "riskpre_*.npz")))


def test_fixtures_present():
    assert len(NAMES) >= 3


@pytest.mark.parametrize("name", NAMES)
def test_turbulence_oracle_matches_reference(name):
    from oracle import riskpre
    z = np.load(os.path.join(GOLDEN, f"riskpre_{name}.npz"), allow_pickle=False)
    got = riskpre.calculate_turbulence(z["close"])
    ref = z["turbulence"]
    np.testing.assert_array_equal(got == 0, ref == 0)
    np.testing.assert_allclose(got, ref, rtol=1e-9)
    assert (ref[:252] == 0).all() and (ref > 0).sum() >= 5


@pytest.mark.parametrize("name", NAMES)
def test_rolling_covariance_oracle_matches_reference(name):
    from oracle import riskpre
    z = np.load(os.path.join(GOLDEN, f"riskpre_{name}.npz"), allow_pickle=False)
    cov = riskpre.rolling_covariance(z["close"], int(z["lookback"]))
    scale = np.abs(z["cov"]).max()
    np.testing.assert_allclose(cov[z["cov_index"]], z["cov"], rtol=0, atol=1e-12 * scale)


def test_short_panel_raises():
    from oracle import riskpre
    with pytest.raises(ValueError):
        riskpre.calculate_turbulence(np.ones((100, 3)))
