"""Device-resident batch of the reference's multi-crypto env
(finrl/meta/env_cryptocurrency_trading/env_multiple_crypto.py:10-111), one HIP launch per step
through the C ABI (finenv_crypto_*)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _native as nat
from .spaces import Box


def action_norm_vector(price0):
    """_generate_action_normalizer, :103-111 (host-side, Python's own math.log / pow)."""
    out = []
    for price in np.asarray(price0, dtype=np.float64):
        x = math.floor(math.log(price, 10))
        out.append(1 / ((10) ** x))
    return np.asarray(out) * 10000


class VecCryptoEnv:
    """E parallel CryptoEnv.  Constructor mirrors the reference: ``config`` holds
    ``price_array`` [T,N] and ``tech_array`` [T,W] (float64)."""

    env_name = "MulticryptoEnv-MI355X"
    if_discrete = False
    target_return = 10

    def __init__(self, config, num_envs, *, lookback=1, initial_capital=1e6, buy_cost_pct=1e-3,
                 sell_cost_pct=1e-3, gamma=0.99, auto_reset=True, device="cuda"):
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nat.FinenvError("finrl_amd has no CPU path: device must be a HIP GPU")
        self.price_array = np.ascontiguousarray(config["price_array"], dtype=np.float64)
        self.tech_array = np.ascontiguousarray(config["tech_array"], dtype=np.float64)
        T, N = self.price_array.shape
        W = self.tech_array.shape[1]
        E = int(num_envs)
        self.num_envs = self.env_num = E
        self.crypto_num = self.action_dim = N
        self.lookback = lookback
        self.max_step = T - lookback - 1                                         # :24
        self.obs_dim = 1 + N + W * lookback
        self.state_dim = 1 + (N + W) * lookback                                  # as declared, :40
        self.gamma = gamma
        self.initial_cash = initial_capital
        self.auto_reset = bool(auto_reset)
        self.action_norm_vector = action_norm_vector(self.price_array[0])
        self.observation_space = Box(-3000, 3000, (self.obs_dim,), np.float32)
        self.action_space = Box(-1, 1, (N,), np.float32)
        L = nat.lib()
        self._cfg = nat.CryptoConfig(E, N, W, T, lookback, 0, float(initial_capital),
                                     float(buy_cost_pct), float(sell_cost_pct), float(gamma))
        self._h = C.c_void_p()
        nat.check(L.finenv_crypto_create(C.byref(self._cfg), C.byref(self._h)), None,
                  "finenv_crypto_create")
        dev = self.device
        self._price = torch.from_numpy(self.price_array).to(dev)
        self._tech = torch.from_numpy((self.tech_array * 2 ** -15).astype(np.float32)).to(dev)
        self._norm = torch.from_numpy(np.ascontiguousarray(self.action_norm_vector)).to(dev)
        self._f64 = torch.zeros(len(nat.CRYPTO_F64_FIELDS), E, dtype=torch.float64, device=dev)
        self._i32 = torch.zeros(len(nat.CRYPTO_I32_FIELDS), E, dtype=torch.int32, device=dev)
        self._stocks = torch.zeros(N, E, dtype=torch.float32, device=dev)
        self.state = {k: self._f64[j] for j, k in enumerate(nat.CRYPTO_F64_FIELDS)}
        self.state.update({k: self._i32[j] for j, k in enumerate(nat.CRYPTO_I32_FIELDS)})
        self.state["stocks"] = self._stocks
        self.state["cash"].fill_(float(initial_capital))                         # __init__ :26-35
        self.state["total_asset"].fill_(float(initial_capital))
        self.state["time"].fill_(lookback - 1)
        pp = nat.CryptoPanelPtrs(self._price.data_ptr(), self._tech.data_ptr(),
                                 self._norm.data_ptr())
        sp = nat.CryptoStatePtrs(self._f64.data_ptr(), self._i32.data_ptr(),
                                 self._stocks.data_ptr())
        nat.check(L.finenv_crypto_bind(self._h, C.byref(pp), C.byref(sp)), self._h, "bind",
                  "crypto")
        self.obs = torch.zeros(E, self.obs_dim, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(E, dtype=torch.float32, device=dev)
        self.done = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.term_obs = None

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().finenv_crypto_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def close(self):
        pass

    def enable_terminal_obs(self):
        import torch
        if self.term_obs is None:
            self.term_obs = torch.zeros_like(self.obs)
        return self.term_obs

    def reset(self, mask=None):
        import torch
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        nat.check(nat.lib().finenv_crypto_reset(self._h, mptr, C.c_void_p(self.obs.data_ptr()),
                                                self._stream()), self._h, "reset", "crypto")
        return self.obs

    supports_record = True      # step(..., record=...) stores the policy's outputs in the same launch

    def step(self, actions, out=None, record=None):
        """actions f32 [E,N] (NOT modified: the reference scales its input in place, :63-65).
        out=(obs, reward, done) optionally directs the outputs into caller tensors, e.g. slice t
        of rollout buffers [n_steps, E, ...] -- collecting a rollout needs no copy.
        record=(values, log_probs, actions_out, values_out, log_probs_out): also copy this step's
        policy outputs into the rollout tensors, in the same launch (finenv_crypto_step_record;
        contiguous float32, 16-byte aligned, E % 4 == 0 -- else use RolloutBuffer.put)."""
        import torch
        if actions.dtype != torch.float32 or not actions.is_contiguous() or \
                actions.device != self.obs.device:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        obs, rew, done = out if out is not None else (self.obs, self.reward, self.done)
        if record is not None:
            v, lp, a_out, v_out, lp_out = record
            for t_ in (v, lp, a_out, v_out, lp_out):
                if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != self.obs.device:
                    raise ValueError("record tensors must be contiguous float32 on the env's device")
            if a_out.numel() != actions.numel() or v.numel() != self.num_envs or \
                    lp.numel() != self.num_envs or v_out.numel() != self.num_envs or \
                    lp_out.numel() != self.num_envs:
                raise ValueError("record: expected values / log_probs [E] and actions_out [E, N]")
            nat.check(nat.lib().finenv_crypto_step_record(
                self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
                C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
                int(self.auto_reset), C.c_void_p(v.data_ptr()), C.c_void_p(lp.data_ptr()),
                C.c_void_p(a_out.data_ptr()), C.c_void_p(v_out.data_ptr()),
                C.c_void_p(lp_out.data_ptr()), self._stream()), self._h, "step_record", "crypto")
            return obs, rew, done, None
        nat.check(nat.lib().finenv_crypto_step(
            self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
            C.c_void_p(self.term_obs.data_ptr()) if self.term_obs is not None else None,
            int(self.auto_reset), self._stream()), self._h, "step", "crypto")
        return obs, rew, done, None

    def as_sb3_vec_env(self):
        """stable-baselines3 VecEnv-shaped view (numpy in / out, auto-reset, terminal_observation)."""
        from .vec_env import SB3VecEnvAdapter
        return SB3VecEnvAdapter(self)

    def episode_return(self):
        """total_asset / initial cash of each env's last finished episode (:89), f32."""
        import torch
        return self.state["episode_return"].to(torch.float32)

    def state_numpy(self):
        out = {k: v.detach().cpu().numpy() for k, v in self.state.items()}
        out["stocks"] = np.ascontiguousarray(out["stocks"].T)
        return out
