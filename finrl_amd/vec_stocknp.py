"""Device-resident batch of the reference's array-state StockTradingEnv
(finrl/meta/env_stock_trading/env_stocktrading_np.py:8-169; ElegantRL / RLlib-facing),
one HIP launch per step through the C ABI (finenv_stocknp_*)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as nat
from .spaces import Box

TAG_PY, TAG_F32, TAG_F64 = 0, 1, 2


def derive_arrays(price_array, tech_array, turbulence_array, turbulence_thresh=99):
    """Host-side array preparation of __init__ (:27-35) and sigmoid_sign (:164-169)."""
    price = np.asarray(price_array).astype(np.float32)
    tech = np.asarray(tech_array).astype(np.float32)
    tech = tech * 2 ** -7
    turb = np.asarray(turbulence_array)
    turb_bool = (turb > turbulence_thresh).astype(np.float32)

    def sigmoid(x):
        return 1 / (1 + np.exp(-x * np.e)) - 0.5
    turb_ary = (sigmoid(turb / turbulence_thresh) * turbulence_thresh * 2 ** -5).astype(np.float32)
    return price, tech, turb_ary, turb_bool


class VecStockTradingEnvNP:
    """E parallel copies; constructor mirrors the reference (``config`` dict with
    price_array / tech_array / turbulence_array / if_train)."""

    env_name = "StockEnv-MI355X"
    if_discrete = False
    target_return = 10.0

    def __init__(self, config, num_envs, *, gamma=0.99, turbulence_thresh=99, min_stock_rate=0.1,
                 max_stock=1e2, initial_capital=1e6, buy_cost_pct=1e-3, sell_cost_pct=1e-3,
                 reward_scaling=2 ** -11, initial_stocks=None, auto_reset=True, device="cuda",
                 seed=0, obs_amount_floor=0.0, obs_pitch=None):
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
        price, tech, turb_ary, turb_bool = derive_arrays(
            config["price_array"], config["tech_array"], config["turbulence_array"],
            turbulence_thresh)
        self.price_ary, self.tech_ary = price, tech
        self.turbulence_ary, self.turbulence_bool = turb_ary, turb_bool
        self.if_train = bool(config.get("if_train", False))
        T, N = price.shape
        W = tech.shape[1]
        E = int(num_envs)
        self.num_envs = self.env_num = E
        self.action_dim = N
        self.state_dim = 1 + 2 + 3 * N + W                                        # :63
        self.max_step = T - 1                                                     # :67
        self.gamma, self.max_stock, self.initial_capital = gamma, max_stock, initial_capital
        self.auto_reset = bool(auto_reset)
        self.observation_space = Box(-3000, 3000, (self.state_dim,), np.float32)
        self.action_space = Box(-1, 1, (N,), np.float32)
        self.initial_stocks = np.zeros(N, np.float32) if initial_stocks is None else \
            np.asarray(initial_stocks, np.float32)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        L = nat.lib()
        self._cfg = nat.StockNpConfig(E, N, W, T, int(max_stock * min_stock_rate), 0,
                                      float(max_stock), float(buy_cost_pct), float(sell_cost_pct),
                                      float(reward_scaling), float(gamma), float(obs_amount_floor))
        self._h = C.c_void_p()
        nat.check(L.finenv_stocknp_create(C.byref(self._cfg), C.byref(self._h)), None,
                  "finenv_stocknp_create")
        D = self.state_dim
        tmpl = np.zeros((T, D), np.float32)
        tmpl[:, 1] = turb_ary
        tmpl[:, 2] = turb_bool
        tmpl[:, 3:3 + N] = price * np.array(2 ** -6, dtype=np.float32)           # :151, :157
        tmpl[:, 3 + 3 * N:] = tech
        dev = self.device
        self._price = torch.from_numpy(np.ascontiguousarray(price)).to(dev)
        self._tmpl = torch.from_numpy(tmpl).to(dev)
        self._tbool = torch.from_numpy(np.ascontiguousarray(turb_bool)).to(dev)
        self._f64 = torch.zeros(len(nat.STOCKNP_F64_FIELDS), E, dtype=torch.float64, device=dev)
        self._i32 = torch.zeros(len(nat.STOCKNP_I32_FIELDS), E, dtype=torch.int32, device=dev)
        self._f32 = torch.zeros(3 * N, E, dtype=torch.float32, device=dev)
        self.state = {k: self._f64[j] for j, k in enumerate(nat.STOCKNP_F64_FIELDS)}
        self.state.update({k: self._i32[j] for j, k in enumerate(nat.STOCKNP_I32_FIELDS)})
        self.state["stocks"] = self._f32[0:N]
        self.state["cool_down"] = self._f32[N:2 * N]
        self.state["stocks0"] = self._f32[2 * N:3 * N]
        self.set_start_state(self.initial_stocks, float(initial_capital), TAG_PY)
        pp = nat.StockNpPanelPtrs(self._price.data_ptr(), self._tmpl.data_ptr(),
                                  self._tbool.data_ptr())
        sp = nat.StockNpStatePtrs(self._f64.data_ptr(), self._i32.data_ptr(),
                                  self._f32.data_ptr())
        nat.check(L.finenv_stocknp_bind(self._h, C.byref(pp), C.byref(sp)), self._h, "bind",
                  "stocknp")
        # obs: [E, D] view of a buffer whose rows start on 64-byte boundaries (see
        # VecStockTradingEnv: packed rows share 64-byte segments that are then written twice)
        if obs_pitch is None:
            obs_pitch = os.environ.get("FINENV_OBS_PITCH", "aligned")
        pitch = (D + 15) // 16 * 16 if obs_pitch == "aligned" else (D if obs_pitch == "packed" else int(obs_pitch))
        if pitch < D:
            raise ValueError("obs_pitch must be >= the observation dimension")
        self._obs_buf = torch.zeros(E, pitch, dtype=torch.float32, device=dev)
        self.obs = self._obs_buf[:, :D]
        self._pitch = self._pitch_set = pitch
        nat.check(L.finenv_stocknp_set_obs_pitch(self._h, pitch), self._h, "set_obs_pitch", "stocknp")
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None

    def set_start_state(self, stocks0, amount0, amount0_tag):
        """Per-env state that reset() restores: stocks0 [N] or [E,N], amount0 scalar or [E],
        dtype tag (TAG_PY for the eval-mode Python float, TAG_F32 for train-mode draws)."""
        import torch
        E, N = self.num_envs, self.action_dim
        s = np.broadcast_to(np.asarray(stocks0, np.float32), (E, N))
        self.state["stocks0"].copy_(torch.from_numpy(np.array(s.T, order="C", copy=True)))
        self.state["amount0"].copy_(torch.from_numpy(
            np.array(np.broadcast_to(np.asarray(amount0, np.float64), (E,)), copy=True)))
        self.state["amount0_tag"].copy_(torch.from_numpy(
            np.array(np.broadcast_to(np.asarray(amount0_tag, np.int32), (E,)), copy=True)))

    def _draw_train_start(self):
        """Train-mode start state (:85-92), drawn on device with this env's generator (the
        reference uses the global numpy RNG, so its draws are not reproducible elsewhere)."""
        import torch
        E, N = self.num_envs, self.action_dim
        st = torch.from_numpy(self.initial_stocks).to(self.device)[:, None] + torch.randint(
            0, 64, (N, E), generator=self._gen, device=self.device).to(torch.float32)
        u = torch.rand(E, generator=self._gen, device=self.device, dtype=torch.float64) * 0.1 + 0.95
        amount = (self.initial_capital * u).to(torch.float32) - (st * self._price[0][:, None]).sum(0)
        self.state["stocks0"].copy_(st)
        self.state["amount0"].copy_(amount.to(torch.float64))
        self.state["amount0_tag"].fill_(TAG_F32)

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().finenv_stocknp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def enable_terminal_obs(self):
        import torch
        if self.term_obs is None:
            self.term_obs = torch.zeros(self.num_envs, self.obs.shape[1], dtype=torch.float32,
                                        device=self.device)
        return self.term_obs

    def _use_pitch(self, pitch):
        if pitch != self._pitch_set:
            nat.check(nat.lib().finenv_stocknp_set_obs_pitch(self._h, int(pitch)), self._h,
                      "set_obs_pitch", "stocknp")
            self._pitch_set = pitch

    def reset(self, mask=None):
        import torch
        if self.if_train:
            self._draw_train_start()
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        self._use_pitch(self._pitch)
        nat.check(nat.lib().finenv_stocknp_reset(self._h, mptr, C.c_void_p(self.obs.data_ptr()),
                                                 self._stream()), self._h, "reset", "stocknp")
        return self.obs

    def step(self, actions, out=None):
        import torch
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        obs, rew, done = out if out is not None else (self.obs, self.reward, self.done)
        if out is not None:
            from .vec_env import _checked_out_pitch
            self._use_pitch(_checked_out_pitch(out, self.obs, self.reward, self.done))
        else:
            self._use_pitch(self._pitch)
        nat.check(nat.lib().finenv_stocknp_step(
            self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
            C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
            int(self.auto_reset), self._stream()), self._h, "step", "stocknp")
        return obs, rew, done, None

    def as_sb3_vec_env(self):
        """stable-baselines3 VecEnv-shaped view (numpy in / out, auto-reset, terminal_observation)."""
        from .vec_env import SB3VecEnvAdapter
        return SB3VecEnvAdapter(self)

    def episode_return(self):
        """total_asset / initial_total_asset of each env's last finished episode (:145), f32."""
        import torch
        return self.state["episode_return"].to(torch.float32)

    def state_numpy(self):
        out = {k: v.detach().cpu().numpy() for k, v in self.state.items()}
        for k in ("stocks", "cool_down", "stocks0"):
            out[k] = np.ascontiguousarray(out[k].T)
        t = out["tags"]
        out["amount_tag"], out["ta_tag"], out["g_tag"] = t & 3, (t >> 2) & 3, (t >> 4) & 3
        out["reward_tag"] = (t >> 8) & 3
        return out
