"""Degenerate panel sizes for the stock env: a one-day panel (every step is terminal, :221),
two- and three-day panels (episodes of 1-2 trading steps), with 1, 5 and 32 tickers, auto-reset
on.  HIP (through the C ABI) vs the oracle, exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("N", [1, 5, 32])
def test_tiny_panels_match_oracle(T, N):
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    E = 70
    rng = np.random.default_rng(T * 10 + N)
    close = (50 + rng.uniform(0, 10, (T, N))).astype(np.float32).astype(np.float64)
    tech = rng.normal(0, 1, (T, 2, N)).astype(np.float32).astype(np.float64)
    risk = np.abs(rng.normal(0, 30, T))
    kw = dict(bench.ENV_KW)
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **kw)
    env.enable_terminal_obs()
    orc = StockOracle(close, tech, risk, n_envs=E, **kw)
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset().astype(np.float32))
    n_done = 0
    for s in range(9):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done, err_msg=f"step {s}")
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        if o_done.any():
            n_done += 1
            np.testing.assert_array_equal(env.term_obs.cpu().numpy()[o_done],
                                          o_term[o_done].astype(np.float32))
    assert n_done >= 3
    st, os_ = env.state_numpy(), orc.state()
    np.testing.assert_array_equal(st["cash"], os_["cash"])
    np.testing.assert_array_equal(st["shares"], os_["shares"])


def test_refresh_after_in_place_state_edit():
    """Editing day / price_day of a batch that already HOLDS shares (what bench.py --desync does
    after a reset) makes the carried begin asset stale; finenv_stock_refresh re-evaluates it.  Every
    env then behaves like a reference env constructed on its own start day (no reset before the
    first step): compared against one oracle per env."""
    _need_gpu()
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    E, T, N, K = 70, 24, 5, 2
    rng = np.random.default_rng(4)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.02, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    risk = np.abs(rng.normal(0, 30, T))
    shares0 = rng.integers(1, 40, N)
    kw = dict(hmax=20, initial_amount=20_000, num_stock_shares=shares0)
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, auto_reset=False, **kw)
    offs = rng.integers(0, T - 6, E).astype(np.int32)
    for k in ("day", "price_day", "start_day"):
        env.state[k].copy_(torch.from_numpy(offs).cuda())
    stale = env.state["begin_asset"].cpu().numpy().copy()
    env.refresh()
    fresh = env.state["begin_asset"].cpu().numpy()
    assert (stale != fresh).any()
    orcs = [StockOracle(close, tech, risk, n_envs=1, day=int(d), **kw) for d in offs]
    np.testing.assert_array_equal(
        env.observe().cpu().numpy(),
        np.stack([np.concatenate([[20_000.0], close[d], shares0, tech[d].reshape(-1)])
                  for d in offs]).astype(np.float32))
    for s in range(5):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        exp = [o.step(a[e:e + 1]) for e, o in enumerate(orcs)]
        np.testing.assert_array_equal(obs.cpu().numpy(),
                                      np.concatenate([x[0] for x in exp]).astype(np.float32))
        np.testing.assert_array_equal(rew.cpu().numpy(),
                                      np.concatenate([x[1] for x in exp]).astype(np.float32),
                                      err_msg=f"reward step {s}")
    np.testing.assert_array_equal(env.state_numpy()["cash"],
                                  np.array([o.state()["cash"][0] for o in orcs]))


def test_step_out_tensors_are_validated():
    """step(out=...) rejects what the kernel would silently mis-write: wrong dtype / device / shape,
    overlapping rows; a single-env batch takes any row stride torch reports."""
    _need_gpu()
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from finrl_amd.vec_stocknp import VecStockTradingEnvNP
    T, N, K = 6, 3, 1
    rng = np.random.default_rng(1)
    close = 100 + rng.random((T, N))
    tech = rng.normal(0, 1, (T, K, N))
    for E in (4, 1):
        env = VecStockTradingEnv(StockPanel(close, tech, np.zeros(T)), E)
        env.reset()
        D = env.state_dim
        a = torch.zeros(E, N, device="cuda")
        rew, done = torch.zeros(E, device="cuda"), torch.zeros(E, dtype=torch.uint8, device="cuda")
        good = torch.zeros(3, E, D, device="cuda")[1]
        obs, _, _, _ = env.step(a, out=(good, rew, done))
        assert torch.equal(obs, env.observe())
        with pytest.raises(ValueError):
            env.step(a, out=(torch.zeros(E, D, dtype=torch.float64, device="cuda"), rew, done))
        with pytest.raises(ValueError):
            env.step(a, out=(torch.zeros(E, D), rew, done))
        with pytest.raises(ValueError):
            env.step(a, out=(good, rew.double(), done))
        with pytest.raises(ValueError):
            env.step(a, out=(torch.zeros(E, D + 1, device="cuda"), rew, done))
        if E > 1:
            with pytest.raises(ValueError):
                env.step(a, out=(torch.zeros(1, D, device="cuda").expand(E, D), rew, done))
    npenv = VecStockTradingEnvNP({"price_array": close, "tech_array": tech.reshape(T, -1),
                                  "turbulence_array": np.zeros(T), "if_train": False}, 4)
    npenv.reset()
    with pytest.raises(ValueError):
        npenv.step(torch.zeros(4, N, device="cuda"),
                   out=(torch.zeros(4, npenv.obs.shape[1], dtype=torch.float64, device="cuda"),
                        npenv.reward, npenv.done))

