"""Shared plumbing of the single-env gym facades: one-env batch on the GPU, numpy in / out."""
from __future__ import annotations

import numpy as np


def to_action_tensor(vec, actions):
    import torch
    a = np.asarray(actions, dtype=np.float32).reshape(1, -1)
    return torch.from_numpy(np.ascontiguousarray(a)).to(vec.device)
