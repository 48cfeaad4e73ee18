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

    def put(self, t, actions, values, log_probs):
        """Store the policy's outputs of step t (one launch: finenv_rollout_put)."""
        import torch
        L = nat.lib()
        L.finenv_rollout_put.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p]
        dev = self.rewards.device
        a = actions.to(dev, torch.float32).contiguous()
        v = values.to(dev, torch.float32).contiguous()
        lp = log_probs.to(dev, torch.float32).contiguous()
        if a.numel() != self.actions[t].numel() or v.numel() != self.num_envs or lp.numel() != self.num_envs:
            raise ValueError("put: expected actions [E, A], values [E], log_probs [E]")
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nat.check(L.finenv_rollout_put(
            C.c_void_p(a.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(lp.data_ptr()),
            C.c_void_p(self.actions[t].data_ptr()), C.c_void_p(self.values[t].data_ptr()),
            C.c_void_p(self.log_probs[t].data_ptr()), self.num_envs,
            self.actions.shape[2], stream), None, "finenv_rollout_put")

    def step(self, env, t, actions, values, log_probs):
        """Store the policy's outputs of step t and step the env into slice t: one launch where the
        env can record them itself (`supports_record`: finenv_crypto_step_record), else two."""
        import torch
        out = (self.obs[t + 1], self.rewards[t], self.dones[t])
        dev, E, A = self.rewards.device, self.num_envs, self.actions.shape[2]
        # anything the one-launch form cannot take (other device, dtype, layout, size, alignment)
        # goes through put(), which converts, + a plain step
        fused = getattr(env, "supports_record", False) and E % 4 == 0 and (E * A) % 4 == 0 and \
            all(torch.is_tensor(x) and x.device == dev and x.dtype == torch.float32 and
                x.is_contiguous() and x.data_ptr() % 16 == 0 and x.numel() == n
                for x, n in ((actions, E * A), (values, E), (log_probs, E)))
        if fused:
            env.step(actions, out=out, record=(values, log_probs, self.actions[t], self.values[t],
                                               self.log_probs[t]))
        else:
            self.put(t, actions, values, log_probs)
            env.step(self.actions[t], out=out)

    def collect(self, env, policy, first_obs):
        """policy(obs) -> (actions [E,A] f32, values [E], log_probs [E]) on device."""
        self.obs[0].copy_(first_obs)
        for t in range(self.n_steps):
            a, v, lp = policy(self.obs[t])
            self.step(env, t, a, v, lp)
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
