"""HIP CryptoEnv (through the C ABI) vs the reference fixtures and the CPU oracle: float32
observations / stocks and float64 cash / assets / rewards compared for exact equality."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[len("crypto_"):-4]
               for p in glob.glob(os.path.join(GOLDEN, "crypto_*.npz")))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")


@pytest.mark.parametrize("name", NAMES)
def test_crypto_hip_matches_reference_fixture(name):
    _need_gpu()
    from finrl_amd.vec_crypto import VecCryptoEnv
    z = np.load(os.path.join(GOLDEN, f"crypto_{name}.npz"), allow_pickle=False)
    T, N, W, S, L = z["cfg_int"].tolist()
    cap, bc, sc, g = z["cfg_float"].tolist()
    E = 70
    env = VecCryptoEnv({"price_array": z["price"], "tech_array": z["tech"]}, E, lookback=L,
                       initial_capital=cap, buy_cost_pct=bc, sell_cost_pct=sc, gamma=g,
                       auto_reset=False)
    np.testing.assert_array_equal(env.action_norm_vector, z["norm"])
    resets = dict(zip(z["reset_step"].tolist(), z["reset_obs"]))
    obs = env.reset().cpu().numpy()
    np.testing.assert_array_equal(obs, np.broadcast_to(resets[-1], obs.shape))
    nd = 0
    for s in range(S):
        a = torch.from_numpy(np.broadcast_to(z["actions"][s], (E, N)).copy()).cuda()
        obs, rew, done, _ = env.step(a)
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        st = env.state_numpy()
        for e in (0, 63, 64, E - 1):
            assert bool(done[e]) == bool(z["done"][s]) and st["time"][e] == z["time"][s], (s, e)
            np.testing.assert_array_equal(st["stocks"][e], z["stocks"][s], err_msg=f"step {s}")
            assert st["cash"][e] == z["cash"][s], (s, e)
            assert st["total_asset"][e] == z["total_asset"][s], (s, e)
            assert st["gamma_return"][e] == z["gamma_return"][s], (s, e)
            assert st["last_reward"][e] == z["reward"][s], (s, e)
            assert rew[e] == np.float32(z["reward"][s])
            np.testing.assert_array_equal(obs[e], z["obs"][s], err_msg=f"obs step {s}")
        if z["done"][s]:
            nd += 1
            obs = env.reset().cpu().numpy()
            np.testing.assert_array_equal(obs, np.broadcast_to(resets[s], obs.shape))
    assert nd == 2


@pytest.mark.parametrize("cfg", [dict(E=1000, T=40, N=10, W=40, L=1, steps=90),
                                 dict(E=130, T=25, N=3, W=7, L=3, steps=60),
                                 dict(E=65, T=16, N=32, W=1, L=2, steps=40),
                                 dict(E=64, T=12, N=1, W=0, L=1, steps=30),
                                 # padded-width boundaries (8 / 16 / 32-wide builds) and the
                                 # observation block path: D = 64 (even), D = 65 (row-wise path)
                                 dict(E=200, T=20, N=17, W=5, L=1, steps=45),
                                 dict(E=128, T=20, N=16, W=47, L=1, steps=45),
                                 dict(E=70, T=20, N=8, W=56, L=1, steps=45),
                                 # more than 64 indicator columns: the trader + streamer kernel
                                 # without the column split (whole rows from the env wave)
                                 dict(E=192, T=20, N=10, W=40, L=2, steps=45),
                                 dict(E=256, T=20, N=20, W=33, L=3, steps=45)])
def test_crypto_hip_matches_oracle_random_batch(cfg):
    _need_gpu()
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    E, T, N, W, L = cfg["E"], cfg["T"], cfg["N"], cfg["W"], cfg["L"]
    rng = np.random.default_rng(E + N)
    price = 10.0 ** rng.uniform(-1, 4.5, N) * np.exp(
        np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    kw = dict(lookback=L, initial_capital=3e4, buy_cost_pct=0.0012, sell_cost_pct=0.0008,
              gamma=0.98)
    orc = CryptoOracle(price, tech, n_envs=E, **kw)
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E, **kw)
    env.enable_terminal_obs()
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset())
    nd = 0
    for s in range(cfg["steps"]):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        a[rng.random((E, N)) < 0.1] = 0.0
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done)
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs, err_msg=f"obs step {s}")
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        st, os_ = env.state_numpy(), orc.state()
        for k in ("cash", "total_asset", "gamma_return", "episode_return", "time", "stocks"):
            np.testing.assert_array_equal(st[k], os_[k], err_msg=f"{k} step {s}")
        if o_done.any():
            nd += 1
            np.testing.assert_array_equal(env.term_obs.cpu().numpy()[o_done], o_term[o_done])
    assert nd >= 2


def test_crypto_rollout_slices_need_no_copy():
    """Outputs written straight into slice t of [n_steps, E, ...] rollout tensors."""
    _need_gpu()
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    E, T, N, W, n_steps = 256, 50, 10, 40, 8
    rng = np.random.default_rng(3)
    price = 100 * np.exp(np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
    orc = CryptoOracle(price, tech, n_envs=E)
    env.reset(); orc.reset()
    obs_buf = torch.zeros(n_steps, E, env.obs_dim, device="cuda")
    rew_buf = torch.zeros(n_steps, E, device="cuda")
    done_buf = torch.zeros(n_steps, E, dtype=torch.uint8, device="cuda")
    acts = rng.uniform(-1, 1, (n_steps, E, N)).astype(np.float32)
    for t in range(n_steps):
        env.step(torch.from_numpy(acts[t]).cuda(), out=(obs_buf[t], rew_buf[t], done_buf[t]))
    for t in range(n_steps):
        o_obs, o_rew, o_done, _ = orc.vec_step(acts[t])
        np.testing.assert_array_equal(obs_buf[t].cpu().numpy(), o_obs)
        np.testing.assert_array_equal(rew_buf[t].cpu().numpy(), o_rew.astype(np.float32))


def test_rollout_buffer_and_gae_scan():
    """Rollout collection straight into [n_steps, E, ...] tensors + GAE scan kernel vs the
    NumPy restatement of SB3's documented formula (float32, exact)."""
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    from oracle.gae import gae
    E, T, N, W, n_steps = 300, 20, 10, 40, 40
    rng = np.random.default_rng(5)
    price = 100 * np.exp(np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
    orc = CryptoOracle(price, tech, n_envs=E)
    buf = RolloutBuffer(n_steps, E, env.obs_dim, N)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)

    outs = []

    def policy(obs):
        a = torch.rand(E, N, generator=gen, device="cuda") * 2 - 1
        outs.append((a, obs[:, 0] * 0.5 + a.sum(1) * 0.01, -a.abs().sum(1)))
        return outs[-1]

    first = env.reset()
    orc.reset()
    last_obs = buf.collect(env, policy, first)
    for t, (a, v, lp) in enumerate(outs):         # the one-launch store of the policy's outputs
        assert torch.equal(buf.actions[t], a) and torch.equal(buf.values[t], v)
        assert torch.equal(buf.log_probs[t], lp)
    acts = buf.actions.cpu().numpy()
    for t in range(n_steps):                      # crosses two episode boundaries (auto-reset)
        o_obs, o_rew, o_done, _ = orc.vec_step(acts[t])
        np.testing.assert_array_equal(buf.obs[t + 1].cpu().numpy(), o_obs)
        np.testing.assert_array_equal(buf.rewards[t].cpu().numpy(), o_rew.astype(np.float32))
        np.testing.assert_array_equal(buf.dones[t].cpu().numpy().astype(bool), o_done)
    assert buf.dones.sum().item() >= 2 * E
    last_values = last_obs[:, 0] * 0.5
    adv, ret = buf.compute_returns_and_advantage(last_values, 0.99, 0.95)
    e_adv, e_ret = gae(buf.rewards.cpu().numpy(), buf.values.cpu().numpy(),
                       buf.dones.cpu().numpy(), last_values.cpu().numpy(), 0.99, 0.95)
    np.testing.assert_array_equal(adv.cpu().numpy(), e_adv)
    np.testing.assert_array_equal(ret.cpu().numpy(), e_ret)


def test_rollout_put_odd_sizes():
    """finenv_rollout_put on sizes / offsets that rule out 16-byte accesses (scalar path)."""
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    E, A = 301, 3
    buf = RolloutBuffer(5, E, 7, A)
    for t in range(5):
        a, v, lp = torch.randn(E, A, device="cuda"), torch.randn(E, device="cuda"), torch.randn(E, device="cuda")
        buf.put(t, a, v, lp)
        assert torch.equal(buf.actions[t], a) and torch.equal(buf.values[t], v)
        assert torch.equal(buf.log_probs[t], lp)
    with pytest.raises(ValueError):
        buf.put(0, torch.zeros(E, A + 1, device="cuda"), torch.zeros(E, device="cuda"),
                torch.zeros(E, device="cuda"))


def test_crypto_step_record_matches_put_plus_step():
    """finenv_crypto_step_record: the policy's outputs land in the rollout tensors and the step's
    results are those of the plain step (two envs side by side); odd sizes fall back to two launches."""
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_crypto import VecCryptoEnv
    rng = np.random.default_rng(8)
    T, N, W = 30, 10, 40
    price = 100 * np.exp(np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    tech = rng.normal(0, 3000, (T, W))
    for E in (1024, 300, 301):           # 301: E % 4 != 0 -> RolloutBuffer.step takes the two-launch path
        a_env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
        b_env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
        buf = RolloutBuffer(6, E, a_env.obs_dim, N)
        a_env.reset()
        b_env.reset()
        for t in range(6):
            a = torch.rand(E, N, device="cuda") * 2 - 1
            v, lp = torch.randn(E, device="cuda"), torch.randn(E, device="cuda")
            buf.step(a_env, t, a, v, lp)
            obs, rew, done, _ = b_env.step(a)
            assert torch.equal(buf.actions[t], a) and torch.equal(buf.values[t], v)
            assert torch.equal(buf.log_probs[t], lp)
            assert torch.equal(buf.obs[t + 1], obs) and torch.equal(buf.rewards[t], rew)
            assert torch.equal(buf.dones[t], done)
    big = VecCryptoEnv({"price_array": price, "tech_array": tech}, 140_000)   # four-wave blocks
    buf = RolloutBuffer(2, 140_000, big.obs_dim, N)
    big.reset()
    a = torch.rand(140_000, N, device="cuda") * 2 - 1
    v, lp = torch.randn(140_000, device="cuda"), torch.randn(140_000, device="cuda")
    buf.step(big, 0, a, v, lp)
    assert torch.equal(buf.actions[0], a) and torch.equal(buf.values[0], v) and torch.equal(buf.log_probs[0], lp)


def _crypto_panel(rng, T, N, W):
    price = 10.0 ** rng.uniform(-1, 4.5, N) * np.exp(
        np.cumsum(rng.normal(0, 0.004, (T, N)), axis=0))
    return price, rng.normal(0, 3000, (T, W))


def _assert_crypto_state_equal(env, orc, tag):
    st, os_ = env.state_numpy(), orc.state()
    for k in ("cash", "total_asset", "gamma_return", "episode_return", "time", "stocks"):
        np.testing.assert_array_equal(st[k], os_[k], err_msg=f"{k} {tag}")


@pytest.mark.parametrize("record", [False, True])
def test_crypto_large_batch_regime_matches_oracle(record):
    """E = 140,000 > 131,072 envs: the launch shape of the large-batch regime (the 262,144-env line
    of bench.py runs it).  Every observation, reward, done flag, terminal observation and state
    field of every env against the oracle over two episode ends; plain step and step_record."""
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    E, T, N, W, steps = 140_000, 12, 10, 40, 24
    rng = np.random.default_rng(77)
    price, tech = _crypto_panel(rng, T, N, W)
    kw = dict(initial_capital=2e5, buy_cost_pct=0.0012, sell_cost_pct=0.0008, gamma=0.97)
    orc = CryptoOracle(price, tech, n_envs=E, **kw)
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E, **kw)
    env.enable_terminal_obs()
    buf = RolloutBuffer(steps, E, env.obs_dim, N) if record else None
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset())
    nd = 0
    for s in range(steps):
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        a[rng.random((E, N)) < 0.1] = 0.0
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        at = torch.from_numpy(a).cuda()
        if record:
            v, lp = torch.randn(E, device="cuda"), torch.randn(E, device="cuda")
            buf.step(env, s, at, v, lp)
            g_obs, g_rew, g_done = buf.obs[s + 1], buf.rewards[s], buf.dones[s]
            assert torch.equal(buf.actions[s], at) and torch.equal(buf.values[s], v)
            assert torch.equal(buf.log_probs[s], lp)
        else:
            g_obs, g_rew, g_done, _ = env.step(at)
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done)
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs, err_msg=f"obs step {s}")
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        _assert_crypto_state_equal(env, orc, f"step {s}")
        if o_done.any():
            nd += 1
            np.testing.assert_array_equal(env.term_obs.cpu().numpy()[o_done], o_term[o_done])
    assert nd >= 2


@pytest.mark.parametrize("cfg", [dict(E=1000, T=30, N=10, W=40, L=1),      # column split (streamer)
                                 dict(E=333, T=30, N=10, W=40, L=2),       # 80 indicator columns
                                 dict(E=140_000, T=16, N=10, W=40, L=1)])  # large-batch regime
def test_crypto_masked_reset_desynchronises_envs(cfg):
    """finenv_crypto_reset(mask) on a random subset in the middle of an episode, then keep stepping:
    the envs of one wave sit on different `time` values (per-env indicator rows in the streamer, the
    generic row writer in the env wave), episodes end at different steps."""
    _need_gpu()
    from finrl_amd.vec_crypto import VecCryptoEnv
    from oracle.crypto import CryptoOracle
    E, T, N, W, L = cfg["E"], cfg["T"], cfg["N"], cfg["W"], cfg["L"]
    rng = np.random.default_rng(E + L)
    price, tech = _crypto_panel(rng, T, N, W)
    kw = dict(lookback=L, initial_capital=5e4, gamma=0.95)
    orc = CryptoOracle(price, tech, n_envs=E, **kw)
    env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E, **kw)
    env.enable_terminal_obs()
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset())
    saw_partial_done = False
    for s in range(2 * T + 6):
        if s in (3, 7, 8, T + 5):          # reset 30 % / 50 % / 2 % / 30 % of the envs
            frac = {3: 0.3, 7: 0.5, 8: 0.02, T + 5: 0.3}[s]
            mask = rng.random(E) < frac
            o_rows = orc.reset_masked(mask)
            g_rows = env.reset(torch.from_numpy(mask.astype(np.uint8))).cpu().numpy()
            np.testing.assert_array_equal(g_rows[mask], o_rows[mask], err_msg=f"reset at {s}")
            _assert_crypto_state_equal(env, orc, f"after reset at {s}")
        a = rng.uniform(-1, 1, (E, N)).astype(np.float32)
        o_obs, o_rew, o_done, o_term = orc.vec_step(a)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(g_done.cpu().numpy().astype(bool), o_done)
        np.testing.assert_array_equal(g_obs.cpu().numpy(), o_obs, err_msg=f"obs step {s}")
        np.testing.assert_array_equal(g_rew.cpu().numpy(), o_rew.astype(np.float32))
        _assert_crypto_state_equal(env, orc, f"step {s}")
        if o_done.any():
            saw_partial_done |= not o_done.all()
            np.testing.assert_array_equal(env.term_obs.cpu().numpy()[o_done], o_term[o_done])
    assert saw_partial_done and len(np.unique(orc.state()["time"])) > 1


def test_rollout_step_takes_any_policy_output_placement():
    """RolloutBuffer.step with policy outputs the one-launch form cannot take (host tensors, float64,
    a transposed view) and E % 4 == 0: falls back to put() + step() instead of handing a host
    pointer to the kernel; results equal the all-device path."""
    _need_gpu()
    from finrl_amd.rollout import RolloutBuffer
    from finrl_amd.vec_crypto import VecCryptoEnv
    rng = np.random.default_rng(12)
    T, N, W, E = 20, 10, 40, 256
    price, tech = _crypto_panel(rng, T, N, W)
    a_env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
    b_env = VecCryptoEnv({"price_array": price, "tech_array": tech}, E)
    buf_a, buf_b = RolloutBuffer(4, E, a_env.obs_dim, N), RolloutBuffer(4, E, b_env.obs_dim, N)
    a_env.reset(); b_env.reset()
    for t in range(4):
        a = torch.rand(E, N) * 2 - 1
        v, lp = torch.randn(E), torch.randn(E)
        forms = [(a, v, lp),                                               # host tensors
                 (a.cuda(), v.double().cuda(), lp.cuda()),                 # float64 values
                 (a.cuda().t().contiguous().t(), v.cuda(), lp.cuda()),     # non-contiguous actions
                 (a.cuda(), v.cuda(), lp)][t]                              # one host tensor
        buf_a.step(a_env, t, *forms)
        buf_b.step(b_env, t, a.cuda(), v.cuda(), lp.cuda())
        for k in ("actions", "values", "log_probs", "rewards", "dones"):
            assert torch.equal(getattr(buf_a, k)[t], getattr(buf_b, k)[t]), (k, t)
        assert torch.equal(buf_a.obs[t + 1], buf_b.obs[t + 1])
