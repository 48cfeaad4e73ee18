"""oracle/cashpenalty_oracle.c vs the committed outputs of the unmodified reference
StockTradingEnvCashpenalty (tests/golden/cashpenalty_*.npz).

done / date_index / market part of the observation: exact.  Money (cash, reward) and holdings:
rtol 1e-12 -- the reference sums its three dot products per step with BLAS ddot, whose
accumulation order is unspecified; the restatement sums left to right."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("cashpenalty_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "cashpenalty_*.npz")))


def make_oracle(z, n_envs=1):
    from oracle.cashpenalty import CashPenaltyOracle
    T, N, Cc, S, disc, inc, use_t, patient = z["cfg_int"].tolist()
    hmax, bc, sc, init, prop, thr = z["cfg_float"].tolist()
    return CashPenaltyOracle(z["close"], z["info"], z["turb"], n_envs=n_envs, buy_cost_pct=bc,
                             sell_cost_pct=sc, hmax=hmax, discrete_actions=bool(disc),
                             shares_increment=inc,
                             turbulence_threshold=thr if use_t else None, initial_amount=init,
                             cash_penalty_proportion=prop, patient=bool(patient))


@pytest.mark.parametrize("name", NAMES)
def test_cashpenalty_oracle_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, f"cashpenalty_{name}.npz"), allow_pickle=False)
    T, N, Cc, S = z["cfg_int"].tolist()[:4]
    o = make_oracle(z)
    ri = 0
    obs = o.reset(z["reset_start"][ri])
    np.testing.assert_allclose(obs[0], z["reset_obs"][ri], rtol=1e-12)
    ri += 1
    nd = 0
    for s in range(S):
        obs, rew, done = o.step(z["actions"][s])
        st = o.state()
        assert done[0] == z["done"][s] and st["date_index"][0] == z["date_index"][s], s
        np.testing.assert_allclose(st["holdings"][0], z["holdings"][s], rtol=1e-12, atol=1e-12,
                                   err_msg=f"holdings {s}")
        assert st["coh"][0] == pytest.approx(z["coh"][s], rel=1e-12), s
        assert rew[0] == pytest.approx(z["reward"][s], rel=1e-10, abs=1e-15), s
        np.testing.assert_array_equal(obs[0][1 + N:], z["obs"][s][1 + N:])
        np.testing.assert_allclose(obs[0][:1 + N], z["obs"][s][:1 + N], rtol=1e-12, atol=1e-12)
        assert st["sum_trades"][0] == pytest.approx(z["sum_trades"][s], rel=1e-6)
        if done[0]:
            nd += 1
            obs = o.reset(z["reset_start"][ri])
            np.testing.assert_allclose(obs[0], z["reset_obs"][ri], rtol=1e-12)
            ri += 1
    assert nd >= 2


def test_upstream_zero_step_restated():
    """tests/environments/test_cash_penalty.py:29-52 on synthetic data (the upstream fixture
    downloads prices): zero actions => cash == initial, no holdings, step counter advances."""
    z = np.load(os.path.join(GOLDEN, "cashpenalty_continuous.npz"), allow_pickle=False)
    o = make_oracle(z)
    o.reset(0)
    N = z["close"].shape[1]
    for i in range(2):
        obs, rew, done = o.step(np.zeros((1, N), np.float32))
        st = o.state()
        assert obs[0][0] == z["cfg_float"][3] and st["logged_total"][0] == z["cfg_float"][3]
        assert obs[0][1:1 + N].sum() == 0
        assert st["date_index"][0] - st["start"][0] == i + 1


def test_upstream_patient_restated():
    """test_cash_penalty.py:55-75: patient=True and an unaffordable buy => nothing bought,
    episode continues."""
    from oracle.cashpenalty import CashPenaltyOracle
    z = np.load(os.path.join(GOLDEN, "cashpenalty_patient.npz"), allow_pickle=False)
    first_close = z["close"][0, 0]
    o = CashPenaltyOracle(z["close"], z["info"], z["turb"], initial_amount=first_close,
                          hmax=first_close * 100, patient=True)
    o.reset(0)
    obs, rew, done = o.step(np.ones((1, z["close"].shape[1]), np.float32))
    assert not done[0] and obs[0][1:1 + z["close"].shape[1]].sum() == 0
