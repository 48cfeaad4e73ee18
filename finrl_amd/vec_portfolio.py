"""Device-resident batch of StockPortfolioEnv instances (env_portfolio.py:15-261 in the
reference tree), one HIP launch per step through the C ABI (finenv_portfolio_*)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from .panel import PortfolioPanel
from .spaces import Box


class VecStockPortfolioEnv:
    """E parallel StockPortfolioEnv.  step(actions f32 [E,N]) -> (obs f32 [E, N+K, N] flattened
    to [E, D], reward f32 [E] = new portfolio value (:196), done u8 [E], None)."""

    if_discrete = False
    env_name = "StockPortfolioEnv-MI355X"

    def __init__(self, panel: PortfolioPanel, num_envs: int, *, initial_amount=1_000_000,
                 auto_reset=True, device="cuda"):
        import torch
        self.panel = panel
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
        E, N, K, T = int(num_envs), panel.N, panel.K, panel.T
        self.num_envs = self.env_num = E
        self.stock_dim = self.action_dim = N
        self.state_dim = panel.D
        self.max_step = T - 1
        self.auto_reset = bool(auto_reset)
        self.observation_space = Box(-np.inf, np.inf, (N + K, N), np.float32)   # :99-103
        self.action_space = Box(0.0, 1.0, (N,), np.float32)                      # :96
        L = nat.lib()
        self._cfg = nat.PortfolioConfig(E, N, K, T, float(initial_amount))
        self._h = C.c_void_p()
        nat.check(L.finenv_portfolio_create(C.byref(self._cfg), C.byref(self._h)), None,
                  "finenv_portfolio_create")
        dev = self.device
        self._f64 = torch.zeros(len(nat.PORTFOLIO_F64_FIELDS), E, dtype=torch.float64, device=dev)
        self._i32 = torch.zeros(len(nat.PORTFOLIO_I32_FIELDS), E, dtype=torch.int32, device=dev)
        self.state = {k: self._f64[j] for j, k in enumerate(nat.PORTFOLIO_F64_FIELDS)}
        self.state.update({k: self._i32[j] for j, k in enumerate(nat.PORTFOLIO_I32_FIELDS)})
        self.state["value"].fill_(float(initial_amount))
        self._panel_t = panel.to_device(dev)
        pp = nat.PortfolioPanelPtrs(self._panel_t["gross_ret"].data_ptr(),
                                    self._panel_t["obs_tmpl"].data_ptr())
        sp = nat.PortfolioStatePtrs(self._f64.data_ptr(), self._i32.data_ptr())
        nat.check(L.finenv_portfolio_bind(self._h, C.byref(pp), C.byref(sp)), self._h, "bind",
                  "portfolio")
        self.obs = torch.zeros(E, panel.D, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None
        self.weights = None

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().finenv_portfolio_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def enable_terminal_obs(self):
        import torch
        if self.term_obs is None:
            self.term_obs = torch.zeros_like(self.obs)
        return self.term_obs

    def enable_weights(self):
        import torch
        if self.weights is None:
            self.weights = torch.zeros(self.num_envs, self.stock_dim, dtype=torch.float32,
                                       device=self.device)
        return self.weights

    def reset(self, mask=None):
        import torch
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        nat.check(nat.lib().finenv_portfolio_reset(self._h, mptr, C.c_void_p(self.obs.data_ptr()),
                                                   self._stream()), self._h, "reset", "portfolio")
        return self.obs

    def step(self, actions, out=None):
        import torch
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        obs, rew, done = out if out is not None else (self.obs, self.reward, self.done)
        nat.check(nat.lib().finenv_portfolio_step(
            self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
            C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
            C.c_void_p(self.weights.data_ptr()) if self.weights is not None else None,
            int(self.auto_reset), self._stream()), self._h, "step", "portfolio")
        return obs, rew, done, None

    def as_sb3_vec_env(self):
        """stable-baselines3 VecEnv-shaped view (numpy in / out, auto-reset, terminal_observation)."""
        from .vec_env import SB3VecEnvAdapter
        return SB3VecEnvAdapter(self)

    def episode_return(self):
        """portfolio value / initial amount per env, f32 (the quantity gathered across ranks)."""
        import torch
        return (self.state["value"] / float(self._cfg.initial_amount)).to(torch.float32)

    def state_numpy(self):
        return {k: v.detach().cpu().numpy() for k, v in self.state.items()}
