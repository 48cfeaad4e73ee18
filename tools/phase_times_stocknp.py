#!/usr/bin/env python3
"""Per-phase timeline of the array-state stock env's step kernel (diagnostic library only)."""
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
    from finrl_amd.vec_stocknp import VecStockTradingEnvNP
    T, N, K = bench.N_DAYS, bench.N_TICKERS, bench.N_TECH
    close, tech, risk = bench.synth_panel()
    env = VecStockTradingEnvNP({"price_array": close, "tech_array": tech.transpose(0, 2, 1).reshape(T, N * K),
                                "turbulence_array": risk * 2, "if_train": False}, E)
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
    names = {0: "trader start", 1: "tile + state staged", 2: "sells + buys done", 3: "asset / reward",
             4: "heads filled", 5: "head chunks written", 6: "state stored",
             11: "  (actions / prices in registers)", 7: "  (sells done)", 12: "  (reciprocals done)", 8: "  (buys done)",
             9: "streamer starts storing", 10: "streamer done"}
    print(f"stocknp E={E} waves={nw}; us since the first wave started (median; p95)")
    for k, n in names.items():
        v = rel[:, :, k].reshape(-1)
        print(f"  {k:2d} {n:28s} {np.median(v):7.2f}  [{np.percentile(v, 95):7.2f}]")

if __name__ == "__main__":
    main()
