"""ctypes binding of libfinenv.so (include/finenv.h).  Fails loudly: no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FINENV_LIB overrides the library path (diagnostic builds, e.g. libfinenv_diag.so)
LIB_PATH = os.environ.get("FINENV_LIB") or os.path.join(_HERE, "lib", "libfinenv.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

FINENV_OK = 0
ABI_VERSION = 3            # include/finenv.h FINENV_ABI_VERSION
AUDIT_HEAD = 4             # FINENV_AUDIT_HEAD: begin cash, asset value, reward, flags
AUDIT_F_LAST_DATE, AUDIT_F_CASH_SHORTAGE, AUDIT_F_TURBULENCE, AUDIT_F_STOP_LOSS, \
    AUDIT_F_LOW_PROFIT, AUDIT_F_HIGH_PROFIT = 1, 2, 4, 8, 16, 32


class NativeLibraryError(ImportError):
    pass


class FinenvError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC_DIR, "-s"] + (["-B"] if force else [])
    subprocess.check_call(args)
    return LIB_PATH


class StockConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_tech", C.c_int32),
        ("n_days", C.c_int32), ("hmax", C.c_int32), ("use_turbulence", C.c_int32),
        ("reset_quirk", C.c_int32), ("initial", C.c_int32), ("track_stats", C.c_int32),
        ("single_ticker", C.c_int32),
        ("buy_cost_pct", C.c_double), ("sell_cost_pct", C.c_double),
        ("reward_scaling", C.c_double), ("turbulence_threshold", C.c_double),
    ]


class StockPanelPtrs(C.Structure):
    _fields_ = [("close", C.c_void_p), ("obs_tmpl", C.c_void_p), ("risk", C.c_void_p)]


# Field order of the two [field][E] state blocks (include/finenv.h enums)
STOCK_F64_FIELDS = ("cash", "cost", "last_reward", "turbulence", "asset0", "ret_sum", "ret_sumsq",
                    "cash0", "begin_asset")
STOCK_I32_FIELDS = ("day", "price_day", "trades", "episode", "start_day")


class StockStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]


class PortfolioConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_tech", C.c_int32),
                ("n_days", C.c_int32), ("initial_amount", C.c_double)]


class PortfolioPanelPtrs(C.Structure):
    _fields_ = [("gross_ret", C.c_void_p), ("obs_tmpl", C.c_void_p)]


PORTFOLIO_F64_FIELDS = ("value", "last_reward")
PORTFOLIO_I32_FIELDS = ("day",)


class PortfolioStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]


class CryptoConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_assets", C.c_int32), ("n_tech", C.c_int32),
                ("n_steps", C.c_int32), ("lookback", C.c_int32), ("reserved0", C.c_int32),
                ("initial_cash", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("gamma", C.c_double)]


class CryptoPanelPtrs(C.Structure):
    _fields_ = [("price", C.c_void_p), ("tech_scaled", C.c_void_p), ("norm", C.c_void_p)]


CRYPTO_F64_FIELDS = ("cash", "total_asset", "gamma_return", "episode_return", "last_reward")
CRYPTO_I32_FIELDS = ("time",)


class CryptoStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p), ("stocks", C.c_void_p)]


class StockNpConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_tickers", C.c_int32), ("n_techw", C.c_int32),
                ("n_days", C.c_int32), ("min_action", C.c_int32), ("reserved0", C.c_int32),
                ("max_stock", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("reward_scaling", C.c_double),
                ("gamma", C.c_double), ("obs_amount_floor", C.c_double)]


class StockNpPanelPtrs(C.Structure):
    _fields_ = [("price", C.c_void_p), ("obs_tmpl", C.c_void_p), ("turb_bool", C.c_void_p)]


STOCKNP_F64_FIELDS = ("amount", "total_asset", "gamma_reward", "initial_total_asset",
                      "episode_return", "last_reward", "amount0")
STOCKNP_I32_FIELDS = ("day", "tags", "amount0_tag")


class StockNpStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p), ("f32", C.c_void_p)]


class CashPenaltyConfig(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("n_assets", C.c_int32), ("n_cols", C.c_int32),
                ("n_days", C.c_int32), ("discrete_actions", C.c_int32),
                ("shares_increment", C.c_int32), ("use_turbulence", C.c_int32),
                ("patient", C.c_int32), ("hmax", C.c_double), ("buy_cost_pct", C.c_double),
                ("sell_cost_pct", C.c_double), ("initial_amount", C.c_double),
                ("cash_penalty_proportion", C.c_double), ("turbulence_threshold", C.c_double)]


class CashPenaltyPanelPtrs(C.Structure):
    _fields_ = [("close", C.c_void_p), ("info", C.c_void_p), ("turb", C.c_void_p)]


CASHPENALTY_F64_FIELDS = ("coh", "turbulence", "sum_trades", "logged_total", "logged_cash")
CASHPENALTY_I32_FIELDS = ("date_index", "start", "episode", "next_start")


class CashPenaltyStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]


class StopLossConfig(C.Structure):
    _fields_ = CashPenaltyConfig._fields_ + [("stoploss_penalty", C.c_double),
                                             ("min_profit_penalty", C.c_double)]


class StopLossPanelPtrs(C.Structure):
    _fields_ = [("close", C.c_void_p), ("info", C.c_void_p), ("turb", C.c_void_p)]


STOPLOSS_F64_FIELDS = ("coh", "turbulence", "sum_trades", "logged_total", "logged_cash",
                       "actual_num_trades")
STOPLOSS_BOOKS = ("holdings", "prev_holdings", "closing_diff_avg_buy",
                  "profit_sell_diff_avg_buy", "n_buys", "avg_buy_price")
STOPLOSS_I32_FIELDS = ("date_index", "start", "episode", "next_start")


class StopLossStatePtrs(C.Structure):
    _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C finrl_amd/csrc` (needs hipcc). finrl_amd has no CPU "
            "fallback.")
    L = C.CDLL(LIB_PATH)
    L.finenv_abi_version.restype = C.c_int
    L.finenv_strerror.restype = C.c_char_p
    L.finenv_strerror.argtypes = [C.c_int]
    L.finenv_device_count.restype = C.c_int
    L.finenv_stock_create.argtypes = [C.POINTER(StockConfig), C.POINTER(C.c_void_p)]
    L.finenv_stock_destroy.argtypes = [C.c_void_p]
    L.finenv_stock_destroy.restype = None
    L.finenv_stock_last_error.argtypes = [C.c_void_p]
    L.finenv_stock_last_error.restype = C.c_char_p
    L.finenv_stock_obs_dim.argtypes = [C.c_void_p]
    L.finenv_stock_set_obs_pitch.argtypes = [C.c_void_p, C.c_int32]
    L.finenv_stock_set_desync_hint.argtypes = [C.c_void_p, C.c_int32]
    L.finenv_stock_bind.argtypes = [C.c_void_p, C.POINTER(StockPanelPtrs),
                                    C.POINTER(StockStatePtrs)]
    L.finenv_stock_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_stock_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_stock_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_stock_refresh.argtypes = [C.c_void_p, C.c_void_p]
    L.finenv_stock_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_stock_episode_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_portfolio_create.argtypes = [C.POINTER(PortfolioConfig), C.POINTER(C.c_void_p)]
    L.finenv_portfolio_destroy.argtypes = [C.c_void_p]
    L.finenv_portfolio_destroy.restype = None
    L.finenv_portfolio_last_error.argtypes = [C.c_void_p]
    L.finenv_portfolio_last_error.restype = C.c_char_p
    L.finenv_portfolio_obs_dim.argtypes = [C.c_void_p]
    L.finenv_portfolio_bind.argtypes = [C.c_void_p, C.POINTER(PortfolioPanelPtrs),
                                        C.POINTER(PortfolioStatePtrs)]
    L.finenv_portfolio_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_portfolio_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p]
    L.finenv_crypto_create.argtypes = [C.POINTER(CryptoConfig), C.POINTER(C.c_void_p)]
    L.finenv_crypto_destroy.argtypes = [C.c_void_p]
    L.finenv_crypto_destroy.restype = None
    L.finenv_crypto_last_error.argtypes = [C.c_void_p]
    L.finenv_crypto_last_error.restype = C.c_char_p
    L.finenv_crypto_obs_dim.argtypes = [C.c_void_p]
    L.finenv_crypto_bind.argtypes = [C.c_void_p, C.POINTER(CryptoPanelPtrs),
                                     C.POINTER(CryptoStatePtrs)]
    L.finenv_crypto_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_crypto_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_crypto_step_record.argtypes = [C.c_void_p] * 6 + [C.c_int32] + [C.c_void_p] * 6
    L.finenv_stocknp_create.argtypes = [C.POINTER(StockNpConfig), C.POINTER(C.c_void_p)]
    L.finenv_stocknp_destroy.argtypes = [C.c_void_p]
    L.finenv_stocknp_destroy.restype = None
    L.finenv_stocknp_last_error.argtypes = [C.c_void_p]
    L.finenv_stocknp_last_error.restype = C.c_char_p
    L.finenv_stocknp_obs_dim.argtypes = [C.c_void_p]
    L.finenv_stocknp_set_obs_pitch.argtypes = [C.c_void_p, C.c_int32]
    L.finenv_stocknp_bind.argtypes = [C.c_void_p, C.POINTER(StockNpPanelPtrs),
                                      C.POINTER(StockNpStatePtrs)]
    L.finenv_stocknp_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_stocknp_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_cashpenalty_create.argtypes = [C.POINTER(CashPenaltyConfig), C.POINTER(C.c_void_p)]
    L.finenv_cashpenalty_destroy.argtypes = [C.c_void_p]
    L.finenv_cashpenalty_destroy.restype = None
    L.finenv_cashpenalty_last_error.argtypes = [C.c_void_p]
    L.finenv_cashpenalty_last_error.restype = C.c_char_p
    L.finenv_cashpenalty_obs_dim.argtypes = [C.c_void_p]
    L.finenv_cashpenalty_bind.argtypes = [C.c_void_p, C.POINTER(CashPenaltyPanelPtrs),
                                          C.POINTER(CashPenaltyStatePtrs)]
    L.finenv_cashpenalty_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_cashpenalty_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_stoploss_create.argtypes = [C.POINTER(StopLossConfig), C.POINTER(C.c_void_p)]
    L.finenv_stoploss_destroy.argtypes = [C.c_void_p]
    L.finenv_stoploss_destroy.restype = None
    L.finenv_stoploss_last_error.argtypes = [C.c_void_p]
    L.finenv_stoploss_last_error.restype = C.c_char_p
    L.finenv_stoploss_obs_dim.argtypes = [C.c_void_p]
    L.finenv_stoploss_bind.argtypes = [C.c_void_p, C.POINTER(StopLossPanelPtrs),
                                       C.POINTER(StopLossStatePtrs)]
    L.finenv_stoploss_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_stoploss_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_riskpre_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.finenv_riskpre_turbulence.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_void_p]
    L.finenv_riskpre_rolling_cov.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_void_p]
    L.finenv_cashpenalty_set_random_start.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
    L.finenv_stoploss_set_random_start.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
    L.finenv_cashpenalty_set_audit.argtypes = [C.c_void_p, C.c_void_p]
    L.finenv_stoploss_set_audit.argtypes = [C.c_void_p, C.c_void_p]
    if L.finenv_abi_version() != ABI_VERSION:
        raise NativeLibraryError("libfinenv.so ABI version mismatch; rebuild (make -C finrl_amd/csrc)")
    L.finenv_struct_size.argtypes = [C.c_int]
    for which, cls in enumerate((StockConfig, StockPanelPtrs, StockStatePtrs, PortfolioConfig,
                                 PortfolioPanelPtrs, PortfolioStatePtrs, CryptoConfig,
                                 CryptoPanelPtrs, CryptoStatePtrs, StockNpConfig,
                                 StockNpPanelPtrs, StockNpStatePtrs, CashPenaltyConfig,
                                 CashPenaltyPanelPtrs, CashPenaltyStatePtrs, StopLossConfig,
                                 StopLossPanelPtrs, StopLossStatePtrs)):
        if L.finenv_struct_size(which) != C.sizeof(cls):
            raise NativeLibraryError(
                f"ABI struct size mismatch for {cls.__name__}: python {C.sizeof(cls)} vs "
                f"library {L.finenv_struct_size(which)}")
    _lib = L
    return L


def check(code: int, handle=None, what: str = "", kind: str = "stock"):
    if code == FINENV_OK:
        return
    L = lib()
    msg = L.finenv_strerror(code).decode()
    if handle:
        detail = getattr(L, f"finenv_{kind}_last_error")(handle).decode()
        if detail:
            msg = f"{msg}: {detail}"
    raise FinenvError(f"{what or 'finenv'} failed ({code}): {msg}")
