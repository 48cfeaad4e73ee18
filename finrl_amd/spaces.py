"""Minimal stand-in for ``gym.spaces.Box`` (gym is an optional dependency; the reference
only uses ``.shape`` / ``.low`` / ``.high`` / ``.dtype``: agents/stablebaselines3/models.py:87,
env_stocktrading.py:60-63).  If gym/gymnasium is importable, ``as_gym()`` converts."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(s) for s in shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else \
            np.asarray(low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else \
            np.asarray(high, dtype=self.dtype)

    def sample(self, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def as_gym(self):
        try:
            from gymnasium import spaces
        except ImportError:
            from gym import spaces
        return spaces.Box(low=self.low, high=self.high, shape=self.shape, dtype=self.dtype.type)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
