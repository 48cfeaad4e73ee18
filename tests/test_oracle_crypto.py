"""oracle/crypto_oracle.c vs the committed outputs of the unmodified reference CryptoEnv
(tests/golden/crypto_*.npz).  Everything is compared for exact equality (float32 obs and
stocks, float64 cash / assets / rewards, done, time, the action normaliser)."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("crypto_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "crypto_*.npz")))


def make_oracle(z, n_envs=1):
    from oracle.crypto import CryptoOracle
    T, N, W, S, L = z["cfg_int"].tolist()
    cap, bc, sc, g = z["cfg_float"].tolist()
    return CryptoOracle(z["price"], z["tech"], n_envs=n_envs, lookback=L, initial_capital=cap,
                        buy_cost_pct=bc, sell_cost_pct=sc, gamma=g)


@pytest.mark.parametrize("name", NAMES)
def test_crypto_oracle_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, f"crypto_{name}.npz"), allow_pickle=False)
    T, N, W, S, L = z["cfg_int"].tolist()
    o = make_oracle(z)
    np.testing.assert_array_equal(o.norm, z["norm"])
    assert o.D == 1 + N + W * L
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    np.testing.assert_array_equal(o.reset()[0], resets[-1])
    nd = 0
    for s in range(S):
        obs, rew, done = o.step(z["actions"][s])
        st = o.state()
        assert done[0] == z["done"][s] and st["time"][0] == z["time"][s], s
        np.testing.assert_array_equal(st["stocks"][0], z["stocks"][s], err_msg=f"stocks {s}")
        assert st["cash"][0] == z["cash"][s], (s, st["cash"][0], z["cash"][s])
        assert st["total_asset"][0] == z["total_asset"][s], s
        assert st["gamma_return"][0] == z["gamma_return"][s], s
        assert rew[0] == z["reward"][s], (s, rew[0], z["reward"][s])
        np.testing.assert_array_equal(obs[0], z["obs"][s], err_msg=f"obs {s}")
        if done[0]:
            nd += 1
            np.testing.assert_array_equal(o.reset()[0], resets[s])
    assert nd == 2
