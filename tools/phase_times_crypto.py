#!/usr/bin/env python3
"""Per-phase timeline of the crypto step kernel from in-kernel s_memrealtime stamps (diagnostic
library only).  Usage: python3 tools/phase_times_crypto.py [E]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FINENV_LIB", os.path.join(ROOT, "finrl_amd", "lib", "libfinenv_diag.so"))

def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    import torch
    from finrl_amd import _native as nat
    from finrl_amd.vec_crypto import VecCryptoEnv
    rng = np.random.default_rng(0)
    T, N, W = 43_200, 10, 40
    price = 10.0 ** rng.uniform(0, 4.5, N) * np.exp(np.cumsum(rng.normal(0, 0.0005, (T, N)), axis=0))
    env = VecCryptoEnv({"price_array": price, "tech_array": rng.normal(0, 3000, (T, W))}, E)
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
    names = ["start", "state + tile staged (round trip 1)", "prices + indicators issued", "sells done",
             "buys done", "asset / reward", "heads in LDS, state stores issued", "obs rows stored",
             "end"]
    print(f"crypto E={E} waves={nw}; us since the first wave started (median over waves and 20 launches; p95)")
    for k, n in enumerate(names):
        v = rel[:, :, k].reshape(-1)
        print(f"  {k} {n:40s} {np.median(v):7.2f}  [{np.percentile(v, 95):7.2f}]")
    print(f"last wave ends at {np.median(rel[:, :, 8].max(axis=1)):.2f} us")

if __name__ == "__main__":
    main()
