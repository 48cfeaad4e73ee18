"""finrl_amd -- MI355X-native batched market environments behind FinRL's gym.Env surface.

Scope: the StockTradingEnv.step/reset hot path of superyuri/FinRL
(finrl/meta/env_stock_trading) and its sibling envs, as hand-written HIP kernels for
gfx950 behind a C ABI (include/finenv.h), with PyTorch-ROCm tensors as device buffers.
There is no CPU fallback: importing the native layer without libfinenv.so, or stepping
without a HIP device, raises.
"""
from .spaces import Box  # noqa: F401
from .panel import StockPanel  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # Lazy: these need the native library.
    if name in ("VecStockTradingEnv", "SB3VecEnvAdapter"):
        from . import vec_env
        return getattr(vec_env, name)
    raise AttributeError(name)
