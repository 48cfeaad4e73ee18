"""Degenerate panel sizes for the stock env: a one-day panel (every step is terminal, :221),
two- and three-day panels (episodes of 1-2 trading steps), with 1, 5 and 32 tickers, auto-reset
on.  HIP (through the C ABI) vs the oracle, exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


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
