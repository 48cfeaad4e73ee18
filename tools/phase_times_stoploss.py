#!/usr/bin/env python3
"""Per-phase timeline of the stop-loss step kernel (diagnostic library only)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FINENV_LIB", os.path.join(ROOT, "finrl_amd", "lib", "libfinenv_diag.so"))

def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    import torch
    import bench
    from finrl_amd import _native as nat
    from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecStopLossEnv
    rng = np.random.default_rng(0)
    T, N, Cc = bench.N_DAYS, bench.N_TICKERS, 5
    close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
    panel = CashPenaltyPanel(close, rng.normal(0, 10, (T, N, Cc)), np.abs(rng.normal(0, 30, T)))
    env = VecStopLossEnv(panel, E, hmax=2_000, random_start=True)
    env.reset()
    nw = (E + 63) // 64
    buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
    pool = [torch.rand(E, N, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(300):
        env.step(pool[i & 7])
    L = nat.lib()
    L.finenv_diag_set_stamp_buffer.argtypes = [C.c_void_p]
    L.finenv_diag_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    acc = []
    for i in range(20):
        buf.zero_()
        env.step(pool[i & 7])
        torch.cuda.synchronize()
        acc.append(buf.cpu().numpy().reshape(nw, 16).astype(np.float64) * 0.01)
    a = np.stack(acc)
    rel = a - a[:, :, 0].min(axis=1)[:, None, None]
    names = {0: "trader: start", 1: "trader: staged (tile, close rows)", 2: "trader: pass 2 + decision",
             3: "trader: its pass 3 done", 4: "trader: past barrier 1", 5: "trader: scalars / terminal rows",
             6: "trader: its half of chunk 0 stored", 8: "streamer: start", 9: "streamer: close rows gathered",
             10: "streamer: market-data chunks stored", 11: "streamer: decision seen",
             12: "streamer: its pass 3 done", 13: "streamer: end"}
    print(f"stoploss E={E} blocks={nw}; us since the first wave started (median; p95)")
    for k, n in names.items():
        v = rel[:, :, k].reshape(-1)
        print(f"  {k:2d} {n:40s} {np.median(v):7.2f}  [{np.percentile(v, 95):7.2f}]")

if __name__ == "__main__":
    main()
