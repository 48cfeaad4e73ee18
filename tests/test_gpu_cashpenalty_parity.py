"""HIP cash-penalty env (through the C ABI) vs the reference fixtures (rtol 1e-12 on money: the
reference's BLAS dot order is unspecified) and vs the CPU oracle (bit-exact: same order)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("cashpenalty_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "cashpenalty_*.npz")))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


def _kw(z):
    T, N, Cc, S, disc, inc, use_t, patient = z["cfg_int"].tolist()
    hmax, bc, sc, init, prop, thr = z["cfg_float"].tolist()
    return dict(buy_cost_pct=bc, sell_cost_pct=sc, hmax=hmax, discrete_actions=bool(disc),
                shares_increment=inc, turbulence_threshold=thr if use_t else None,
                initial_amount=init, cash_penalty_proportion=prop, patient=bool(patient))


@pytest.mark.parametrize("name", NAMES)
def test_cashpenalty_hip_matches_reference_fixture(name):
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
    z = np.load(os.path.join(GOLDEN, f"cashpenalty_{name}.npz"), allow_pickle=False)
    T, N, Cc, S = z["cfg_int"].tolist()[:4]
    E = 70
    env = VecCashPenaltyEnv(CashPenaltyPanel(z["close"], z["info"], z["turb"]), E,
                            random_start=False, auto_reset=False, **_kw(z))
    ri = 0
    env.set_next_start(int(z["reset_start"][ri]))
    obs = env.reset().cpu().numpy()
    np.testing.assert_allclose(obs[0], z["reset_obs"][ri].astype(np.float32), rtol=1e-6)
    ri += 1
    nd = 0
    for s in range(S):
        a = torch.from_numpy(np.broadcast_to(z["actions"][s], (E, N)).copy()).cuda()
        obs, rew, done, _ = env.step(a)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        st = env.state_numpy()
        for e in (0, 63, 64, E - 1):
            assert bool(done[e]) == bool(z["done"][s]), (s, e)
            assert st["date_index"][e] == z["date_index"][s], (s, e)
            np.testing.assert_allclose(st["holdings"][e], z["holdings"][s], rtol=1e-12, atol=1e-12)
            assert st["coh"][e] == pytest.approx(z["coh"][s], rel=1e-12)
            assert rew[e] == pytest.approx(z["reward"][s], rel=1e-6, abs=1e-12)
            np.testing.assert_array_equal(obs[e][1 + N:], z["obs"][s][1 + N:].astype(np.float32))
            np.testing.assert_allclose(obs[e][:1 + N], z["obs"][s][:1 + N].astype(np.float32),
                                       rtol=1e-6, atol=1e-6)
        if z["done"][s]:
            nd += 1
            env.set_next_start(int(z["reset_start"][ri]))
            env.reset()
            ri += 1
    assert nd >= 2


@pytest.mark.parametrize("cfg", [
    dict(E=1000, T=30, N=30, C=5, steps=80, hmax=30_000, thr=50.0, patient=False, disc=False),
    dict(E=130, T=20, N=5, C=2, steps=50, hmax=400_000, thr=None, patient=True, disc=False),
    dict(E=65, T=16, N=32, C=1, steps=40, hmax=20_000, thr=None, patient=False, disc=True),
    dict(E=64, T=10, N=1, C=0, steps=25, hmax=1e6, thr=30.0, patient=False, disc=False),
    # nearly every env runs out of cash within a few steps: many more re-decided rows per block than
    # the streamer patches in registers (kFix), and rows on three observation chunks
    dict(E=200, T=40, N=30, C=5, steps=30, hmax=400_000, thr=None, patient=False, disc=False),
    # 241 columns: the streamer copies the market data as 16-byte quads (rows up to 320 columns)
    dict(E=100, T=20, N=30, C=7, steps=30, hmax=60_000, thr=40.0, patient=False, disc=True),
    # 331 columns: wider than that, the trader writes the observation itself (no streamer)
    dict(E=70, T=16, N=30, C=10, steps=24, hmax=60_000, thr=40.0, patient=False, disc=False)])
def test_cashpenalty_hip_matches_oracle_random_batch(cfg):
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
    from oracle.cashpenalty import CashPenaltyOracle
    E, T, N, Cc = cfg["E"], cfg["T"], cfg["N"], cfg["C"]
    rng = np.random.default_rng(E + N)
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    info = rng.normal(0, 10, (T, N, Cc))
    turb = np.abs(rng.normal(0, 30, T))
    kw = dict(hmax=cfg["hmax"], turbulence_threshold=cfg["thr"], patient=cfg["patient"],
              discrete_actions=cfg["disc"], shares_increment=3, initial_amount=5e5,
              buy_cost_pct=0.002, sell_cost_pct=0.001, cash_penalty_proportion=0.15)
    orc = CashPenaltyOracle(close, info, turb, n_envs=E, **kw)
    env = VecCashPenaltyEnv(CashPenaltyPanel(close, info, turb), E, random_start=False, **kw)
    env.enable_terminal_obs()
    starts = rng.integers(0, T // 2, E).astype(np.int32)
    env.set_next_start(starts)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset(starts).astype(np.float32))
    nd = 0
    for s in range(cfg["steps"]):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        starts = rng.integers(0, T // 2, E).astype(np.int32)
        env.set_next_start(starts)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a, starts)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done, err_msg=f"{s}")
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        st, os_ = env.state_numpy(), orc.state()
        for k in ("coh", "holdings", "date_index", "start", "sum_trades", "logged_total",
                  "logged_cash", "episode"):
            np.testing.assert_array_equal(st[k], os_[k], err_msg=f"{k} step {s}")
        if o_done.any():
            nd += 1
            np.testing.assert_array_equal(env.term_obs.cpu().numpy()[o_done],
                                          o_term[o_done].astype(np.float32))
    assert nd >= 2


def test_random_start_is_drawn_on_device():
    """random_start=True (:134-138): starting points drawn on the device, uniformly in
    [0, int(T * 0.5)), per env and per episode, reproducible from the seed; no host work per step."""
    _need_gpu()
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv
    rng = np.random.default_rng(1)
    T, N, E = 40, 4, 4096
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    panel = CashPenaltyPanel(close, rng.normal(0, 1, (T, N, 2)))
    envs = [VecCashPenaltyEnv(panel, E, hmax=1000, random_start=True, seed=s) for s in (7, 7, 8)]
    starts = []
    for env in envs:
        env.reset()
        st = env.state_numpy()
        assert (st["start"] == st["date_index"]).all()
        assert st["start"].min() >= 0 and st["start"].max() < T // 2
        starts.append(st["start"].copy())
    np.testing.assert_array_equal(starts[0], starts[1])          # same seed, same draws
    assert (starts[0] != starts[2]).mean() > 0.8                 # another seed, other draws
    counts = np.bincount(starts[0], minlength=T // 2)
    assert counts.min() > 0.6 * E / (T // 2) and counts.max() < 1.4 * E / (T // 2)   # ~uniform
    env = envs[0]
    zero = torch.zeros(E, N, device="cuda")
    first = starts[0]
    seen = np.zeros(E, bool)
    for s in range(T):                                           # every env reaches the last date
        _, _, done, _ = env.step(zero)
        d = done.cpu().numpy().astype(bool)
        st = env.state_numpy()
        assert (st["start"][d] == st["date_index"][d]).all()     # auto-reset drew a fresh start
        assert st["start"].max() < T // 2
        seen |= d
    assert seen.all()
    assert (env.state_numpy()["start"] != first).mean() > 0.8    # new episode, new draw
