"""NumPy restatement of stable-baselines3's documented
RolloutBuffer.compute_returns_and_advantage (float32) -- TEST INFRASTRUCTURE.
Parity unpinned: SB3 is not vendored in the reference (setup.py:34-36) nor installed here, so
this follows the documented formula; the HIP kernel is checked against this restatement."""
import numpy as np


def gae(rewards, values, dones, last_values, gamma=0.99, gae_lambda=0.95):
    rewards = np.asarray(rewards, np.float32)
    values = np.asarray(values, np.float32)
    S = rewards.shape[0]
    adv = np.zeros_like(rewards)
    last_gae_lam = np.zeros(rewards.shape[1], np.float32)
    g, gl = np.float32(gamma), np.float32(gamma) * np.float32(gae_lambda)
    for step in reversed(range(S)):
        next_values = np.asarray(last_values, np.float32) if step == S - 1 else values[step + 1]
        nnt = np.float32(1.0) - np.asarray(dones[step], np.float32)
        delta = rewards[step] + g * next_values * nnt - values[step]
        last_gae_lam = delta + gl * nnt * last_gae_lam
        adv[step] = last_gae_lam
    return adv, adv + values
