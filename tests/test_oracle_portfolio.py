"""oracle/portfolio_oracle.c vs the committed outputs of the unmodified reference
StockPortfolioEnv (tests/golden/portfolio_*.npz, made by tests/golden/make_golden.py).

Exact: observations (cov + indicator rows), done, day, reset observations.
float32 softmax weights: <= 4 ulp (NumPy's SIMD expf vs libm expf); portfolio value / reward
(fp64, driven by those weights): rel 1e-6 (north-star bound 1e-5)."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("portfolio_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "portfolio_*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN, f"portfolio_{name}.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", NAMES)
def test_portfolio_oracle_matches_reference(name):
    from oracle.portfolio import PortfolioOracle
    z = load(name)
    T, N, K, S = z["cfg_int"].tolist()
    o = PortfolioOracle(z["close"], z["cov"], z["tech"], n_envs=1,
                        initial_amount=z["cfg_float"][0])
    assert o.D == (N + K) * N
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    np.testing.assert_array_equal(o.reset()[0], resets[-1])
    n_done = 0
    for s in range(S):
        obs, rew, done = o.step(z["actions"][s])
        st = o.state()
        assert done[0] == z["done"][s] and st["day"][0] == z["day"][s], s
        np.testing.assert_array_equal(obs[0], z["obs"][s], err_msg=f"obs step {s}")
        assert rew[0] == pytest.approx(z["reward"][s], rel=1e-6), s
        assert st["value"][0] == pytest.approx(z["value"][s], rel=1e-6), s
        if done[0]:
            n_done += 1
            np.testing.assert_array_equal(o.reset()[0], resets[s])
    assert n_done == 2


def test_softmax_weights_within_a_few_ulp():
    import ctypes as C
    from oracle.portfolio import PortfolioOracle
    from oracle.stock import lib, _p
    z = load("dow30")
    T, N, K, S = z["cfg_int"].tolist()
    o = PortfolioOracle(z["close"], z["cov"], z["tech"], n_envs=1)
    o.reset()
    worst = 0.0
    for s in range(T - 1):
        w = np.empty(N, dtype=np.float32)
        rew = np.empty(1)
        done = np.empty(1, dtype=np.uint8)
        lib().pf_oracle_step_env(o._h, C.c_int(0), _p(np.ascontiguousarray(z["actions"][s])),
                                 None, _p(rew), _p(done), _p(w))
        ref = z["weights"][s]
        worst = max(worst, float(np.max(np.abs(w - ref) / np.spacing(ref))))
    assert worst <= 4.0, worst      # ulps (exp +-1, f32 sum +-1, division +-1)
