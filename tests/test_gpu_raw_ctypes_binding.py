"""The ctypes stub INTEGRATION.md shows a FinRL maintainer (section B), executed as written:
its own struct declarations, the layout guard, create / bind / init / reset / step through the
C ABI with nothing from finrl_amd._native -- and the same numbers as VecStockTradingEnv."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_stub_runs_and_matches():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv

    L = C.CDLL(os.path.join(ROOT, "finrl_amd", "lib", "libfinenv.so"))
    assert L.finenv_abi_version() == 3

    class Cfg(C.Structure):                      # finenv_stock_config
        _fields_ = [(n, C.c_int32) for n in ("n_envs", "n_tickers", "n_tech", "n_days", "hmax",
                    "use_turbulence", "reset_quirk", "initial", "track_stats", "single_ticker")] + \
                   [(n, C.c_double) for n in ("buy_cost_pct", "sell_cost_pct", "reward_scaling",
                    "turbulence_threshold")]

    class Panel(C.Structure):                    # finenv_stock_panel
        _fields_ = [(n, C.c_void_p) for n in ("close", "obs_tmpl", "risk")]

    class State(C.Structure):                    # finenv_stock_state
        _fields_ = [("f64", C.c_void_p), ("i32", C.c_void_p)]

    for i, cls in enumerate((Cfg, Panel, State)):
        assert L.finenv_struct_size(i) == C.sizeof(cls)      # layout guard

    close, tech, risk = bench.synth_panel()
    close, tech, risk = close[:50], tech[:50], risk[:50]
    panel = StockPanel(close, tech, risk)
    E, N, K, T = 200, 30, 8, 50
    D = 1 + 2 * N + K * N
    dev = "cuda"
    f64 = torch.zeros(9, E, dtype=torch.float64, device=dev)          # FINENV_SF_* rows
    i32 = torch.zeros(5 + 2 * N, E, dtype=torch.int32, device=dev)    # FINENV_SI_* rows, holdings, shares0
    f64[7] = 1_000_000.0                                              # FINENV_SF_CASH0
    close_t = torch.from_numpy(panel.signed_close()).to(dev)          # sign bit = untradable flag
    tmpl = torch.from_numpy(panel.obs_template()).to(dev)
    risk_t = torch.from_numpy(panel.risk).to(dev)
    obs = torch.zeros(E, D, dtype=torch.float32, device=dev)
    reward = torch.zeros(E, dtype=torch.float32, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)

    h = C.c_void_p()
    cfg = Cfg(E, N, K, T, 100, 0, 1, 1, 1, 0, 1e-3, 1e-3, 1e-4, 0.0)
    L.finenv_stock_create.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_void_p)]
    assert L.finenv_stock_create(C.byref(cfg), C.byref(h)) == 0
    L.finenv_stock_bind.argtypes = [C.c_void_p, C.POINTER(Panel), C.POINTER(State)]
    assert L.finenv_stock_bind(h, C.byref(Panel(close_t.data_ptr(), tmpl.data_ptr(), risk_t.data_ptr())),
                               C.byref(State(f64.data_ptr(), i32.data_ptr()))) == 0
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.finenv_stock_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.finenv_stock_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.finenv_stock_step.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p]
    L.finenv_stock_last_error.argtypes = [C.c_void_p]
    L.finenv_stock_last_error.restype = C.c_char_p
    L.finenv_stock_destroy.argtypes = [C.c_void_p]
    assert L.finenv_stock_init(h, 0, stream) == 0
    assert L.finenv_stock_reset(h, None, C.c_void_p(obs.data_ptr()), stream) == 0

    ref = VecStockTradingEnv(panel, E, **bench.ENV_KW)
    assert torch.equal(ref.reset(), obs)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    for s in range(60):                                               # crosses an episode end
        actions = torch.rand(E, N, generator=gen, device=dev) * 2 - 1
        rc = L.finenv_stock_step(h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
                                 C.c_void_p(reward.data_ptr()), C.c_void_p(done.data_ptr()),
                                 None, None, 1, stream)
        assert rc == 0, L.finenv_stock_last_error(h).decode()
        r_obs, r_rew, r_done, _ = ref.step(actions)
        assert torch.equal(obs, r_obs) and torch.equal(reward, r_rew) and torch.equal(done, r_done)
    assert torch.equal(f64[0], ref.state["cash"])
    # unbound / invalid use reports an error code and a message, it does not crash
    h2 = C.c_void_p()
    assert L.finenv_stock_create(C.byref(cfg), C.byref(h2)) == 0
    assert L.finenv_stock_reset(h2, None, C.c_void_p(obs.data_ptr()), stream) != 0
    assert b"bind" in L.finenv_stock_last_error(h2)
    L.finenv_stock_destroy(h2)
    L.finenv_stock_destroy(h)
