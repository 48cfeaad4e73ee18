"""Full-size checks at BASELINE.json configs[1] (65,536 envs, DOW30 x 8, T=2893 panel), where
replaying the oracle for every env would take minutes: size-independent properties plus an
exact oracle comparison on a sampled subset of envs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

E = 65_536


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


def _panel():
    import bench
    return bench.synth_panel()


def test_fullsize_properties_and_sampled_oracle():
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    close, tech, risk = _panel()
    T, N = close.shape
    kw = dict(bench.ENV_KW, turbulence_threshold=float(np.percentile(risk, 90)))
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **kw)
    env.enable_realised()
    sample = np.sort(np.random.default_rng(0).choice(E, 192, replace=False))
    # the sampled envs include both ends and wave / block boundaries
    sample[:6] = [0, 63, 64, 127, 128, E - 1]
    sample = np.unique(sample)
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **kw)
    obs = env.reset()
    o_obs = orc.reset()
    np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    close_t = torch.from_numpy(close).cuda()
    prev_asset = env.total_asset().clone()
    for s in range(60):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        day_before = env.state["day"].clone()
        obs, rew, done, _ = env.step(a)
        st = env.state
        # --- properties over all 65,536 envs (device-side, fp64) ---------------------------
        assert int((st["holdings"] < 0).sum()) == 0, "negative holdings"
        assert float(st["cash"].min()) >= 0.0, "cash must stay non-negative (cost-aware buys)"
        assert int(done.sum()) == 0
        assert torch.equal(st["day"], day_before + 1)
        # reward == (asset_after - asset_before) * reward_scaling, assets recomputed here
        asset = st["cash"] + (close_t[st["day"].long()] * st["holdings"].T.double()).sum(1)
        np.testing.assert_allclose((rew.double() / kw["reward_scaling"]).cpu().numpy(),
                                   (asset - prev_asset).cpu().numpy(), rtol=0, atol=2.0)
        prev_asset = asset
        # realised trades never exceed the request: |realised| <= |trunc(a*hmax)|
        want = (a * 100.0).to(torch.int32)
        turb = (st["turbulence"] >= kw["turbulence_threshold"])      # next step's flag
        real = env.realised
        ok = (real.abs() <= want.abs()) | (real <= 0)                # liquidation sells exceed
        assert bool(ok.all())
        # observation layout: cash | close | holdings | tech
        assert torch.equal(obs[:, 0], st["cash"].float())
        assert torch.equal(obs[:, 1 + N:1 + 2 * N], st["holdings"].T.float())
        # --- exact oracle parity on the sampled envs ---------------------------------------
        o_obs, o_rew, o_done, _ = orc.vec_step(a[sample].cpu().numpy())
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32),
                                      err_msg=f"step {s}")
        np.testing.assert_array_equal(rew[sample].cpu().numpy(), o_rew.astype(np.float32))
    os_ = orc.state()
    np.testing.assert_array_equal(st["cash"][sample].cpu().numpy(), os_["cash"])
    np.testing.assert_array_equal(st["holdings"].T[sample].cpu().numpy(), os_["shares"])


def test_fullsize_identical_envs_agree_and_zero_action_is_idempotent():
    """All 65,536 envs fed the same actions must stay bit-identical (lane / wave / block
    invariance); a zero action changes nothing but the day (restated from the reference's
    tests/environments/test_cash_penalty.py:29-52)."""
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    close, tech, risk = _panel()
    N = close.shape[1]
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **bench.ENV_KW)
    env.reset()
    rng = np.random.default_rng(3)
    for s in range(40):
        row = rng.uniform(-1, 1, N).astype(np.float32) if s % 4 else np.zeros(N, np.float32)
        a = torch.from_numpy(row).cuda().expand(E, N).contiguous()
        cash0, hold0 = env.state["cash"].clone(), env.state["holdings"].clone()
        obs, rew, done, _ = env.step(a)
        assert bool((obs == obs[0]).all()) and bool((rew == rew[0]).all())
        assert bool((env.state["cash"] == env.state["cash"][0]).all())
        if s % 4 == 0:
            assert torch.equal(env.state["cash"], cash0)
            assert torch.equal(env.state["holdings"], hold0)


def test_fullsize_episode_rollover_checksum():
    """Run across an episode boundary on a short panel at full batch size: every env resets
    inside the launch; a checksum of per-env checksums matches the oracle-verified sample."""
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    close, tech, risk = _panel()
    close, tech, risk = close[:12], tech[:12], risk[:12]
    N = close.shape[1]
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **bench.ENV_KW)
    env.enable_terminal_obs()
    sample = np.arange(0, E, 997)
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **bench.ENV_KW)
    env.reset(); orc.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    n_done = 0
    for s in range(30):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        obs, rew, done, _ = env.step(a)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a[sample].cpu().numpy())
        assert int(done.sum()) in (0, E)                     # lock-step: all or none
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32))
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), o_done)
        if o_done.all():
            n_done += 1
            np.testing.assert_array_equal(env.term_obs[sample].cpu().numpy(),
                                          o_term.astype(np.float32))
            assert bool((env.state["day"] == 0).all()) and bool((env.state["trades"] == 0).all())
    assert n_done == 2
    assert bool((env.state["episode"] == 3).all())           # initial reset + 2 auto-resets


def test_fullsize_two_full_episodes_sampled_oracle():
    """The bench workload end to end: 65,536 envs, the full T=2893 panel, two whole episodes
    (5,786 steps, two in-launch auto-resets, the cash-bound late-episode regime included); 192
    sampled envs are compared with the oracle at every step (observations, rewards, dones) and
    in their final state and Sharpe sums."""
    _need_gpu()
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    close, tech, risk = _panel()
    T, N = close.shape
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **bench.ENV_KW)
    sample = np.sort(np.random.default_rng(5).choice(E, 192, replace=False))
    sample[:4] = [0, 63, 64, E - 1]
    sample = np.unique(sample)
    idx = torch.from_numpy(sample).cuda()
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **bench.ENV_KW)
    env.reset(); orc.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2893)
    n_done = 0
    for s in range(2 * T):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        obs, rew, done, _ = env.step(a)
        o_obs, o_rew, o_done, _ = orc.vec_step(a[idx].cpu().numpy())
        if not (np.array_equal(obs[idx].cpu().numpy(), o_obs.astype(np.float32)) and
                np.array_equal(rew[idx].cpu().numpy(), o_rew.astype(np.float32)) and
                np.array_equal(done[idx].cpu().numpy().astype(bool), o_done)):
            pytest.fail(f"mismatch at step {s}")
        n_done += int(o_done.all())
    assert n_done == 2
    st, os_ = env.state, orc.state()
    np.testing.assert_array_equal(st["cash"][idx].cpu().numpy(), os_["cash"])
    np.testing.assert_array_equal(st["cost"][idx].cpu().numpy(), os_["cost"])
    np.testing.assert_array_equal(st["holdings"].T[idx].cpu().numpy(), os_["shares"])
    np.testing.assert_array_equal(st["trades"][idx].cpu().numpy(), os_["trades"])
    assert bool((st["episode"] == 3).all())


@pytest.mark.parametrize("hint", [True, False])
def test_fullsize_desynchronised_sampled_oracle(hint):
    """65,536 envs driven out of lock step by masked resets (different days inside every wave,
    episode ends at different steps), with and without the desynchronised-batch hint -- the two
    instantiations of the step kernel -- exact against the oracle on a sampled subset."""
    import ctypes as C
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle, lib, _p
    T, N, K = 24, 30, 8
    rng = np.random.default_rng(21)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    tech[:, 0, :][rng.random((T, N)) < 0.03] = 1.0
    risk = np.abs(rng.normal(0, 30, T))
    kw = dict(hmax=100, initial_amount=300_000, turbulence_threshold=50.0)
    sample = np.unique(np.concatenate([[0, 63, 64, 127, 128, E - 1], rng.choice(E, 200, replace=False)]))
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, auto_reset=True, **kw)
    env.hint_desynchronised(hint)
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **kw)
    orc.reset()
    env.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    for s in range(70):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        o_obs, o_rew, o_done, _ = orc.vec_step(a[sample].cpu().numpy())
        obs, rew, done, _ = env.step(a)
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32), err_msg=f"{s}")
        np.testing.assert_array_equal(rew[sample].cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), o_done)
        if s in (2, 5, 9, 14):
            m = rng.random(E) < 0.3
            for j, e in enumerate(sample):
                if m[e]:
                    row = np.empty(orc.D)
                    lib().stock_oracle_reset_env(orc._h, C.c_int(int(j)), _p(row))
            env.reset(torch.from_numpy(m.astype(np.uint8)).cuda())
    st, os_ = env.state_numpy(), orc.state()
    assert len(np.unique(st["day"])) > 4
    np.testing.assert_array_equal(st["day"][sample], os_["day"])
    np.testing.assert_array_equal(st["cash"][sample], os_["cash"])
    np.testing.assert_array_equal(st["shares"][sample], os_["shares"])


@pytest.mark.parametrize("N,E", [(30, 69_700), (100, 66_000), (50, 40_000)])
def test_batches_larger_than_one_round_of_blocks(N, E):
    """More 64-env groups than fit on the chip at once: the step is issued as several equal launches
    (launch_rounds, finenv_stock_common.h), each with its own first group.  Every env of every round -- the
    last group is a partial wave -- against the oracle on sampled envs, plus size-independent properties
    over the whole batch, across an episode end."""
    _need_gpu()
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    T, K = 9, 2
    rng = np.random.default_rng(N + E)
    close = 100 * np.exp(np.cumsum(rng.normal(0, 0.02, (T, N)), axis=0))
    tech = rng.normal(0, 1, (T, K, N))
    risk = np.abs(rng.normal(0, 30, T))
    kw = dict(hmax=40, initial_amount=60_000, turbulence_threshold=45.0)
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **kw)
    env.enable_terminal_obs()
    sample = np.unique(np.concatenate([[0, 63, 64, E - 1, E - 2, E // 2, E // 2 + 1],
                                       rng.choice(E, 250, replace=False)]))
    # the groups around every round boundary a 2- / 3-round split of this batch could have
    blocks = (E + 63) // 64
    for k in (2, 3):
        chunk = (blocks + k - 1) // k
        for b in range(chunk, blocks, chunk):
            sample = np.union1d(sample, [min(E - 1, 64 * b - 1), min(E - 1, 64 * b), min(E - 1, 64 * b + 63)])
    orc = StockOracle(close, tech, risk, n_envs=len(sample), **kw)
    np.testing.assert_array_equal(env.reset()[sample].cpu().numpy(), orc.reset().astype(np.float32))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(E)
    for s in range(2 * T + 1):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        obs, rew, done, _ = env.step(a)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a[sample].cpu().numpy())
        np.testing.assert_array_equal(obs[sample].cpu().numpy(), o_obs.astype(np.float32), err_msg=f"step {s}")
        np.testing.assert_array_equal(rew[sample].cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(done[sample].cpu().numpy().astype(bool), o_done)
        # lock-step batch: every env is on the same day and finishes together
        assert int(done.sum()) in (0, E) and len(torch.unique(env.state["day"])) == 1
        if o_done.any():
            np.testing.assert_array_equal(env.term_obs[sample].cpu().numpy(), o_term.astype(np.float32))
    st, os_ = env.state_numpy(), orc.state()
    np.testing.assert_array_equal(st["cash"][sample], os_["cash"])
    np.testing.assert_array_equal(st["shares"][sample], os_["shares"])
    assert int((env.state["holdings"] < 0).sum()) == 0 and float(env.state["cash"].min()) >= 0.0
