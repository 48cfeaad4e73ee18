"""Device-resident batch of StockTradingEnv instances (one HIP launch per step).

``VecStockTradingEnv`` is tensor-in / tensor-out (the ElegantRL vectorised-env shape:
``env_num, state_dim, action_dim, max_step, if_discrete, target_return``);
``SB3VecEnvAdapter`` presents the same batch through the stable-baselines3 ``VecEnv``
protocol (numpy in / numpy out, auto-reset with ``info["terminal_observation"]``), which is
what the reference builds with ``DummyVecEnv([lambda: env])`` (env_stocktrading.py:549-552).

All arithmetic happens in finrl_amd/csrc/finenv_stock.hip through the C ABI; this module
only owns the torch tensors that back the state and hands their pointers over.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as nat
from .panel import StockPanel
from .spaces import Box


def _torch():
    import torch
    return torch


def _checked_out_pitch(out, obs, reward, done):
    """Validate step(out=(obs, reward, done)) against the env's own output tensors and return the row
    pitch (floats) of out[0].  The kernel writes float32 rows of D columns `pitch` floats apart."""
    o, r, d = out
    E, D = obs.shape
    for t, ref, what in ((o, obs, "out[0]"), (r, reward, "out[1]"), (d, done, "out[2]")):
        if t.dtype != ref.dtype or t.device != ref.device or tuple(t.shape) != tuple(ref.shape):
            raise ValueError(f"{what} must be {tuple(ref.shape)} {ref.dtype} on {ref.device}")
    if o.stride(-1) != 1 or not r.is_contiguous() or not d.is_contiguous():
        raise ValueError("out[0] needs unit column stride; out[1] / out[2] must be contiguous")
    if E == 1:                  # (torch reports an arbitrary row stride for a single row)
        return D
    if o.stride(0) < D:
        raise ValueError("out[0]: rows overlap (row stride smaller than the observation dimension)")
    return o.stride(0)


class VecStockTradingEnv:
    """E parallel copies of the reference ``StockTradingEnv`` (env_stocktrading.py:19-552).

    Constructor keywords keep the reference's names (``hmax, initial_amount,
    num_stock_shares, buy_cost_pct, sell_cost_pct, reward_scaling, turbulence_threshold,
    day, initial``).  ``initial_amount`` / ``num_stock_shares`` may be per-env
    ([E] / [E, N]) which also covers the ``previous_state`` carry-over (:423-450).
    Costs are scalars, as in this fork (:118, :179).
    """

    if_discrete = False
    env_name = "StockTradingEnv-MI355X"
    target_return = 10.0

    def __init__(self, panel: StockPanel, num_envs: int, *, hmax=100,
                 initial_amount=1_000_000, num_stock_shares=None, buy_cost_pct=1e-3,
                 sell_cost_pct=1e-3, reward_scaling=1e-4, turbulence_threshold=None,
                 day=0, initial=True, reset_quirk=True, track_stats=True, auto_reset=True,
                 device="cuda", obs_pitch=None):
        torch = _torch()
        if not isinstance(buy_cost_pct, (int, float)) or not isinstance(sell_cost_pct, (int, float)):
            # the fork's own env raises TypeError on list costs
            raise TypeError("buy_cost_pct / sell_cost_pct must be scalars in this fork")
        self.panel = panel
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
        E, N, K, T = int(num_envs), panel.N, panel.K, panel.T
        self.num_envs = self.env_num = E
        self.stock_dim = self.action_dim = N
        self.state_dim = self.state_space = panel.D
        self.max_step = T - 1
        self.hmax = int(hmax)
        self.reward_scaling = float(reward_scaling)
        self.turbulence_threshold = turbulence_threshold
        self.auto_reset = bool(auto_reset)
        self.observation_space = Box(-np.inf, np.inf, (panel.D,), np.float32)
        self.action_space = Box(-1.0, 1.0, (N,), np.float32)

        self._cfg = nat.StockConfig(
            E, N, K, T, self.hmax, int(turbulence_threshold is not None), int(bool(reset_quirk)),
            int(bool(initial)), int(bool(track_stats)),
            int(N == 1),      # one ticker in the frame: the reference's single-stock branches (:415-422)
            float(buy_cost_pct),
            float(sell_cost_pct), float(reward_scaling),
            float(turbulence_threshold) if turbulence_threshold is not None else 0.0)
        L = nat.lib()
        self._h = C.c_void_p()
        nat.check(L.finenv_stock_create(C.byref(self._cfg), C.byref(self._h)), None,
                  "finenv_stock_create")

        dev = self.device
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        cash0 = np.broadcast_to(np.asarray(initial_amount, dtype=np.float64), (E,))
        if num_stock_shares is None:
            num_stock_shares = np.zeros(N, dtype=np.int64)
        sh0 = np.broadcast_to(np.asarray(num_stock_shares, dtype=np.int64), (E, N))
        # two [field][E] blocks (include/finenv.h); self.state holds named views into them
        nf, ni = len(nat.STOCK_F64_FIELDS), len(nat.STOCK_I32_FIELDS)
        self._state_f64 = torch.zeros(nf, E, **f64)
        self._state_i32 = torch.zeros(ni + 2 * N, E, **i32)
        self.state = {k: self._state_f64[j] for j, k in enumerate(nat.STOCK_F64_FIELDS)}
        self.state.update({k: self._state_i32[j] for j, k in enumerate(nat.STOCK_I32_FIELDS)})
        self.state["holdings"] = self._state_i32[ni:ni + N]
        self.state["shares0"] = self._state_i32[ni + N:ni + 2 * N]
        self.state["cash0"].copy_(torch.from_numpy(np.array(cash0, dtype=np.float64)))
        self.state["shares0"].copy_(torch.from_numpy(np.ascontiguousarray(sh0.T).astype(np.int32)))
        self._panel_t = panel.to_device(dev)
        pp = nat.StockPanelPtrs(*(self._panel_t[k].data_ptr()
                                  for k in ("close", "obs_tmpl", "risk")))
        sp = nat.StockStatePtrs(self._state_f64.data_ptr(), self._state_i32.data_ptr())
        nat.check(L.finenv_stock_bind(self._h, C.byref(pp), C.byref(sp)), self._h, "bind")

        # Observation rows: `obs` is a [E, D] view of a buffer whose rows start on 64-byte boundaries
        # (pitch = D rounded up to 16 floats) unless obs_pitch="packed" / an explicit pitch is given:
        # packed rows of 4*D bytes share their first and last 64-byte segment with a neighbour row
        # written microseconds apart -- two partial HBM writes instead of one.
        # Values and shape are the reference's; only the row stride differs (obs.stride(0)).
        if obs_pitch is None:
            obs_pitch = os.environ.get("FINENV_OBS_PITCH", "aligned")
        if obs_pitch == "aligned":
            pitch = (panel.D + 15) // 16 * 16
        elif obs_pitch == "packed":
            pitch = panel.D
        else:
            pitch = int(obs_pitch)
            if pitch < panel.D:
                raise ValueError("obs_pitch must be >= the observation dimension")
        self._obs_buf = torch.zeros(E, pitch, dtype=torch.float32, device=dev)
        self.obs = self._obs_buf[:, :panel.D]
        self._pitch = self._pitch_set = pitch
        nat.check(L.finenv_stock_set_obs_pitch(self._h, pitch), self._h, "set_obs_pitch")
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None
        self.realised = None
        self._step_args = None
        self._stats = None
        nat.check(L.finenv_stock_init(self._h, int(day), self._stream()), self._h, "init")

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().finenv_stock_destroy(self._h)
                self._h = None
        except Exception:
            pass

    close = __del__

    def enable_terminal_obs(self):
        torch = _torch()
        if self.term_obs is None:
            self.term_obs = torch.zeros(self.num_envs, self.state_dim, dtype=torch.float32,
                                        device=self.device)
        return self.term_obs

    def hint_desynchronised(self, on=True):
        """Performance hint (results never depend on it): the envs of this batch sit on different
        days -- per-env start days, staggered episode ends.  Selects the step-kernel instantiation
        tuned for per-env panel rows (finenv_stock_set_desync_hint)."""
        nat.check(nat.lib().finenv_stock_set_desync_hint(self._h, int(bool(on))), self._h,
                  "set_desync_hint")

    def _use_pitch(self, pitch):
        if pitch != self._pitch_set:
            nat.check(nat.lib().finenv_stock_set_obs_pitch(self._h, int(pitch)), self._h,
                      "set_obs_pitch")
            self._pitch_set = pitch

    def enable_realised(self):
        torch = _torch()
        if self.realised is None:
            self.realised = torch.zeros(self.num_envs, self.stock_dim, dtype=torch.int32,
                                        device=self.device)
        return self.realised

    # ------------------------------------------------------------------ env protocol
    def reset(self, mask=None):
        """reset() (:359-393) for all envs (or mask[e] != 0) -> obs [E, D] f32 (device)."""
        L = nat.lib()
        mptr = None
        if mask is not None:
            torch = _torch()
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        self._use_pitch(self._pitch)
        nat.check(L.finenv_stock_reset(self._h, mptr, C.c_void_p(self.obs.data_ptr()),
                                       self._stream()), self._h, "reset")
        return self.obs

    def refresh(self):
        """Call after editing ``state["cash"]`` / ``state["holdings"]`` / ``state["price_day"]`` in
        place: re-evaluates the carried begin asset (``state["begin_asset"]``) that ``step`` uses
        for the next reward instead of recomputing it (finenv_stock_refresh)."""
        nat.check(nat.lib().finenv_stock_refresh(self._h, self._stream()), self._h, "refresh")

    def observe(self):
        """render() (:395-396): current observation without stepping."""
        self._use_pitch(self._pitch)
        nat.check(nat.lib().finenv_stock_observe(self._h, C.c_void_p(self.obs.data_ptr()),
                                                 self._stream()), self._h, "observe")
        return self.obs

    def step(self, actions, out=None):
        """One step() (:220-357) for every env, asynchronously on the current stream.

        actions: float32 [E, N] CUDA tensor.  Returns (obs, reward, done, info) where the
        first three are views of persistent device tensors, overwritten by the next call
        (clone them to keep).  No host synchronisation happens here.
        out = (obs [E, D] f32, reward [E] f32, done [E] u8): write there instead (rollout
        buffers: the kernel writes straight into slice t, no staging copy).
        """
        torch = _torch()
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device or \
                tuple(actions.shape) != (self.num_envs, self.stock_dim):
            actions = actions.to(device=self.device, dtype=torch.float32).reshape(
                self.num_envs, self.stock_dim).contiguous()
        # the pointers of the persistent output tensors are cached (the Python side of a launch
        # costs more than half of the ~13 us a step call takes on the host)
        key = (self.term_obs is not None, self.realised is not None)
        if self._step_args is None or self._step_args[0] != key:
            self._step_args = (key, nat.lib().finenv_stock_step, (
                C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                C.c_void_p(self.done.data_ptr()),
                C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
                C.c_void_p(self.realised.data_ptr()) if self.realised is not None else None))
        _, fn, outs = self._step_args
        ret = (self.obs, self.reward, self.done)
        if out is not None:
            ret = out
            self._use_pitch(_checked_out_pitch(out, self.obs, self.reward, self.done))
            outs = (C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                    C.c_void_p(out[2].data_ptr())) + outs[3:]
        elif self._pitch_set != self._pitch:
            self._use_pitch(self._pitch)
        rc = fn(self._h, C.c_void_p(actions.data_ptr()), *outs, int(self.auto_reset),
                self._stream())
        if rc:
            nat.check(rc, self._h, "step")
        return ret[0], ret[1], ret[2], None

    # ------------------------------------------------------------------ introspection
    def episode_stats(self):
        """Terminal-branch summary (:226-264) -> f64 [E, 6] device tensor:
        begin_total_asset, end_total_asset, total_reward, total_cost, total_trades, sharpe."""
        torch = _torch()
        if self._stats is None:
            self._stats = torch.zeros(self.num_envs, 6, dtype=torch.float64, device=self.device)
        nat.check(nat.lib().finenv_stock_episode_stats(
            self._h, C.c_void_p(self._stats.data_ptr()), self._stream()), self._h, "stats")
        return self._stats

    def total_asset(self):
        return self.episode_stats()[:, 1]

    def episode_return(self):
        """end_total_asset / begin_total_asset per env (the quantity gathered across ranks)."""
        st = self.episode_stats()
        return (st[:, 1] / st[:, 0]).to(_torch().float32)

    def state_numpy(self):
        """Host copy of the per-env state (synchronises)."""
        out = {k: v.detach().cpu().numpy() for k, v in self.state.items()}
        out["shares"] = np.ascontiguousarray(out.pop("holdings").T)
        out["shares0"] = np.ascontiguousarray(out["shares0"].T)
        return out

    def as_sb3_vec_env(self):
        return SB3VecEnvAdapter(self)


class SB3VecEnvAdapter:
    """stable-baselines3 ``VecEnv``-shaped view of a VecStockTradingEnv.

    SB3 itself is not vendored in the reference (setup.py:34-36) nor installed here, so this
    follows its documented public behaviour: ``reset() -> float32 [E, D]``;
    ``step_wait() -> (obs f32 [E, D], rewards f32 [E], dones bool [E], infos list[dict])``
    with auto-reset and ``infos[i]["terminal_observation"]``.
    """

    def __init__(self, env):
        """env: any of the batched envs of this package (they share the tensor protocol:
        ``reset()``, ``step(a) -> (obs, reward, done, _)``, ``enable_terminal_obs()``)."""
        self.env = env
        env.auto_reset = True
        env.enable_terminal_obs()
        self.num_envs = env.num_envs
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self._actions = None
        self.render_mode = None

    def reset(self):
        return self.env.reset().cpu().numpy()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        torch = _torch()
        a = self._actions
        if not torch.is_tensor(a):
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        a = a.to(self.env.device, non_blocking=True)
        obs, rew, done, _ = self.env.step(a)
        obs_h = obs.cpu().numpy()
        rew_h = rew.cpu().numpy()
        done_h = done.cpu().numpy().astype(bool)
        infos = [{} for _ in range(self.num_envs)]
        if done_h.any():
            idx = np.nonzero(done_h)[0]
            term = self.env.term_obs[torch.from_numpy(idx).to(self.env.device)].cpu().numpy()
            for j, i in enumerate(idx):
                infos[i]["terminal_observation"] = term[j]
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        pass

    def seed(self, seed=None):
        return [seed] * self.num_envs

    def render(self, mode="human"):
        obs = self.env.observe() if hasattr(self.env, "observe") else self.env.obs
        return obs.cpu().numpy()

    def _indices(self, indices):
        """SB3 ``VecEnv._get_indices``: None -> all envs, int -> [int], else the iterable."""
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, (int, np.integer)):
            return [int(indices)]
        return [int(i) for i in indices]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_attr(self, attr_name, indices=None):
        v = getattr(self.env, attr_name)
        return [v] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """SB3's public signature (the reference calls ``env_method(method_name=...)``,
        agents/stablebaselines3/models.py:120-121).  The batch is one object, so the method
        runs once and its result is repeated per selected env."""
        r = getattr(self.env, method_name)(*method_args, **method_kwargs)
        return [r] * len(self._indices(indices))


class SingleEnvVecAdapter:
    """``DummyVecEnv([lambda: env])``-shaped wrapper around one of the single-env facades
    (what the reference's ``get_sb_env`` returns, env_stocktrading.py:549-552): SB3's documented
    VecEnv behaviour -- observation / reward buffers in the spaces' dtype (float32), auto-reset
    on done with ``infos[0]["terminal_observation"]``, ``env_method(method_name, ...)``,
    ``get_attr / set_attr(..., indices)``, ``seed``, ``env_is_wrapped``.  SB3 is not vendored
    in the reference (setup.py:34-36), so this boundary is parity unpinned upstream; the tests
    drive it with the reference's own caller loop (agents/stablebaselines3/models.py:110-129)."""

    def __init__(self, env):
        self.env = env
        self.envs = [env]
        self.num_envs = 1
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self._actions = None
        self.render_mode = None
        self._obs_dtype = np.dtype(getattr(env.observation_space, "dtype", np.float32))

    def _obs(self, obs):
        return np.asarray(obs, dtype=self._obs_dtype)[None].copy()

    def reset(self):
        return self._obs(self.env.reset())

    def step_async(self, actions):
        self._actions = np.asarray(actions)

    def step_wait(self):
        obs, rew, done, info = self.env.step(self._actions[0])
        info = dict(info) if isinstance(info, dict) else {}
        if done:
            info["terminal_observation"] = np.asarray(obs, dtype=self._obs_dtype)
            obs = self.env.reset()
        return (self._obs(obs), np.asarray([rew], dtype=np.float32),
                np.asarray([done], dtype=bool), [info])

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _indices(self, indices):
        if indices is None:
            return [0]
        if isinstance(indices, (int, np.integer)):
            indices = [indices]
        idx = [int(i) for i in indices]
        if any(i != 0 for i in idx):
            raise IndexError(f"indices {idx}: this VecEnv holds one env")
        return idx

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self.env, method_name)(*method_args, **method_kwargs)
                for _ in self._indices(indices)]

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        for _ in self._indices(indices):
            setattr(self.env, attr_name, value)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def seed(self, seed=None):
        fn = getattr(self.env, "seed", None) or getattr(self.env, "_seed", None)
        return [fn(seed) if fn is not None else None]

    def render(self, mode="human"):
        return self.env.render(mode) if hasattr(self.env, "render") else None

    def close(self):
        pass
