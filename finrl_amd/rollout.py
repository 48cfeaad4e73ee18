"""Device-resident PPO rollout buffers filled by the env kernels without copies
(BASELINE.json configs[4]), plus the GAE scan kernel."""
from __future__ import annotations

import ctypes as C

from . import _native as nat


class RolloutBuffer:
    """[n_steps, E, ...] tensors on the env's device.  ``collect(env, policy)`` steps the env
    n_steps times writing observations / rewards / dones directly into slice t (the C ABI
    takes output pointers per step, so no staging copy exists)."""

    def __init__(self, n_steps, num_envs, obs_dim, action_dim, device="cuda"):
        import torch
        self.n_steps, self.num_envs = int(n_steps), int(num_envs)
        kw = dict(device=device, dtype=torch.float32)
        self.obs = torch.zeros(n_steps + 1, num_envs, obs_dim, **kw)     # obs[t] seen before step t
        self.actions = torch.zeros(n_steps, num_envs, action_dim, **kw)
        self.rewards = torch.zeros(n_steps, num_envs, **kw)
        self.dones = torch.zeros(n_steps, num_envs, device=device, dtype=torch.uint8)
        self.values = torch.zeros(n_steps, num_envs, **kw)
        self.log_probs = torch.zeros(n_steps, num_envs, **kw)
        self.advantages = torch.zeros(n_steps, num_envs, **kw)
        self.returns = torch.zeros(n_steps, num_envs, **kw)

    def collect(self, env, policy, first_obs):
        """policy(obs) -> (actions [E,A] f32, values [E], log_probs [E]) on device."""
        self.obs[0].copy_(first_obs)
        for t in range(self.n_steps):
            a, v, lp = policy(self.obs[t])
            self.actions[t].copy_(a)
            self.values[t].copy_(v)
            self.log_probs[t].copy_(lp)
            env.step(self.actions[t], out=(self.obs[t + 1], self.rewards[t], self.dones[t]))
        return self.obs[self.n_steps]

    def compute_returns_and_advantage(self, last_values, gamma=0.99, gae_lambda=0.95):
        import torch
        L = nat.lib()
        L.finenv_gae_scan.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_float,
                                                          C.c_float, C.c_void_p]
        lv = last_values.to(self.rewards.device, torch.float32).contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(self.rewards.device).cuda_stream)
        nat.check(L.finenv_gae_scan(
            C.c_void_p(self.rewards.data_ptr()), C.c_void_p(self.values.data_ptr()),
            C.c_void_p(self.dones.data_ptr()), C.c_void_p(lv.data_ptr()),
            C.c_void_p(self.advantages.data_ptr()), C.c_void_p(self.returns.data_ptr()),
            self.n_steps, self.num_envs, float(gamma), float(gae_lambda), stream), None,
            "finenv_gae_scan")
        return self.advantages, self.returns
