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


_LAZY = {      # name -> module; lazy because these need the native library
    "VecStockTradingEnv": "vec_env", "SB3VecEnvAdapter": "vec_env", "SingleEnvVecAdapter": "vec_env",
    "VecStockPortfolioEnv": "vec_portfolio", "VecCryptoEnv": "vec_crypto",
    "VecStockTradingEnvNP": "vec_stocknp", "VecCashPenaltyEnv": "vec_cashpenalty",
    "VecStopLossEnv": "vec_cashpenalty", "CashPenaltyPanel": "vec_cashpenalty",
    "RolloutBuffer": "rollout", "GraphedSegment": "graph",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    if name == "PortfolioPanel":
        from .panel import PortfolioPanel
        return PortfolioPanel
    raise AttributeError(name)
