"""oracle/stocknp_oracle.c vs the committed outputs of the unmodified reference array-state
env (env_stocktrading_np.py) recorded under NumPy 2.2.6 (tests/golden/stocknp_*.npz).
Exact equality on values AND on the NumPy scalar dtype of amount / total_asset /
gamma_reward / reward (0 = python float, 1 = float32, 2 = float64)."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("stocknp_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "stocknp_*.npz")))


def make_oracle(z, n_envs=1):
    from oracle.stocknp import StockNpOracle
    cap, ms, bc, sc, g = z["cfg_float"].tolist()
    extra = {}
    if "obs_amount_floor" in z.files:            # StockEnvNAS100 fixtures (env_nas100_wrds.py)
        extra = dict(obs_amount_floor=float(z["obs_amount_floor"]),
                     turbulence_thresh=float(z["turbulence_thresh"]))
    return StockNpOracle(z["price_array"], z["tech_array"], z["turbulence_array"],
                         n_envs=n_envs, gamma=g, max_stock=ms, initial_capital=cap,
                         buy_cost_pct=bc, sell_cost_pct=sc, **extra)


@pytest.mark.parametrize("name", NAMES)
def test_stocknp_oracle_matches_reference(name):
    import ctypes as C
    from oracle.stock import lib, _p
    z = np.load(os.path.join(GOLDEN, f"stocknp_{name}.npz"), allow_pickle=False)
    T, N, K, S, if_train = z["cfg_int"].tolist()
    o = make_oracle(z)
    assert o.D == 3 + 3 * N + N * K == z["obs"].shape[1]
    rs = z["reset_step"].tolist()
    ri = 0

    def do_reset():
        nonlocal ri
        o.set_initial(z["reset_stocks0"][ri], z["reset_amount0"][ri], z["reset_amount0_tag"][ri])
        obs = o.reset()
        np.testing.assert_array_equal(obs[0], z["reset_obs"][ri])
        ri += 1

    do_reset()
    nd = 0
    for s in range(S):
        a = np.ascontiguousarray(z["actions"][s])
        obs = np.empty(o.D, np.float32)
        rew = np.empty(1)
        rtag = np.empty(1, np.int32)
        done = np.empty(1, np.uint8)
        lib().np_oracle_step_env(o._h, C.c_int(0), _p(a), _p(obs), _p(rew), _p(rtag), _p(done))
        st = o.state()
        assert bool(done[0]) == bool(z["done"][s]) and st["day"][0] == z["day"][s], s
        np.testing.assert_array_equal(st["stocks"][0], z["stocks"][s], err_msg=f"stocks {s}")
        np.testing.assert_array_equal(st["cool_down"][0], z["cool_down"][s])
        assert (st["amount"][0], st["amount_tag"][0]) == (z["amount"][s], z["amount_tag"][s]), s
        assert (st["total_asset"][0], st["ta_tag"][0]) == (z["total_asset"][s], z["ta_tag"][s]), s
        assert (st["gamma_reward"][0], st["g_tag"][0]) == (z["gamma_reward"][s], z["g_tag"][s]), s
        assert (rew[0], rtag[0]) == (z["reward"][s], z["reward_tag"][s]), s
        assert st["episode_return"][0] == z["episode_return"][s], s
        np.testing.assert_array_equal(obs, z["obs"][s], err_msg=f"obs {s}")
        if done[0]:
            nd += 1
            assert rs[ri] == s
            do_reset()
    assert nd == 2
