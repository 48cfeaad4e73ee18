#!/usr/bin/env python3
"""Per-phase timeline of the stock_step kernel from in-kernel s_memrealtime stamps
(diagnostic library only: FINENV_LIB=finrl_amd/lib/libfinenv_diag.so).  Shares, not
absolute run time: the stamped build is slower than the product build."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FINENV_LIB", os.path.join(ROOT, "finrl_amd", "lib", "libfinenv_diag.so"))


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    NT = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    desync = len(sys.argv) > 3 and sys.argv[3] == "desync"
    import torch
    import bench
    from finrl_amd import StockPanel, _native as nat
    from finrl_amd.vec_env import VecStockTradingEnv
    close, tech, risk = bench.synth_panel(N=NT)
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, **bench.ENV_KW)
    env.reset()
    if desync:          # every env on its own day, like bench.py --desync
        offs = torch.randint(0, close.shape[0] - 1, (E,), device="cuda").to(torch.int32)
        for k in ("day", "price_day", "start_day"):
            env.state[k].copy_(offs)
        env.hint_desynchronised(True)
    nb = (E + 63) // 64
    buf = torch.zeros(nb * 2 * 16, dtype=torch.int64, device="cuda")
    pool = [torch.rand(E, NT, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(200):
        env.step(pool[i & 7])
    L = nat.lib()
    L.finenv_diag_set_stamp_buffer.argtypes = [C.c_void_p]
    L.finenv_diag_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    acc = []
    for i in range(20):
        buf.zero_()
        env.step(pool[i & 7])
        torch.cuda.synchronize()
        acc.append(buf.cpu().numpy().reshape(nb, 2, 16).astype(np.float64) * 0.01)  # -> us
    a = np.stack(acc)                                   # [rep, block, role, stamp]
    t0 = a[:, :, :, 0].min(axis=(1, 2), keepdims=True)[..., None]
    rel = a - t0
    names_t = ["start", "day/pd loaded", "pre-barrier work done", "barrier passed",
               "begin asset", "sorted", "sells done", "buys done", "end asset", "reward/stats",
               "obs chunk0 written", "state written"]
    names_s = ["start", "day/pd loaded", "tile+prices staged", "barrier passed", "streamed",
               "hand-off chunk written"]
    print(f"E={E} blocks={nb}{' desynchronised days' if desync else ''}; times in us since the first wave of the launch started "
          "(median over blocks and 20 launches; p95 in brackets)")
    print("trader wave:")
    for k, n in enumerate(names_t):
        v = rel[:, :, 0, k].reshape(-1)
        print(f"  {k:2d} {n:24s} {np.median(v):7.2f}  [{np.percentile(v, 95):7.2f}]  p99 {np.percentile(v, 99):6.2f}  max {v.max():6.2f}")
    print("streamer wave:")
    for k, n in enumerate(names_s):
        v = rel[:, :, 1, k].reshape(-1)
        print(f"  {k:2d} {n:24s} {np.median(v):7.2f}  [{np.percentile(v, 95):7.2f}]  p99 {np.percentile(v, 99):6.2f}  max {v.max():6.2f}")
    cyc = (a[:, :, 0, 15] - a[:, :, 0, 14]) / 0.01          # raw shader-clock ticks
    us = a[:, :, 0, 11] - a[:, :, 0, 0]
    print(f"trader shader clock: {np.median(cyc / us) / 1e3:.2f} GHz (s_memtime ticks / s_memrealtime us)")
    fin = rel[:, :, 0, 11]
    thr = np.percentile(fin, 98)
    slow = np.argwhere(fin >= thr)[:, 1]
    print("slowest 2 % of traders by block index mod 8:", np.bincount(slow % 8, minlength=8).tolist(),
          " by index quartile:", np.bincount(slow * 4 // nb, minlength=4).tolist())
    sl = fin >= thr
    for k in (1, 2, 7, 9, 10):
        print(f"   slow blocks, trader stamp {k}: median {np.median(rel[:, :, 0, k][sl]):.2f}")
    # blocks b and b + 8 share an XCD (round-robin dispatch): finish times per b % 8
    bi = np.arange(nb) % 8
    print("median finish per block index mod 8 (trader | streamer's last market-data store issued):")
    print("   trader  ", " ".join(f"{np.median(rel[:, bi == x, 0, 11]):6.2f}" for x in range(8)))
    print("   streamer", " ".join(f"{np.median(rel[:, bi == x, 1, 4]):6.2f}" for x in range(8)))
    print("   trader loads back (stamp 1)", " ".join(f"{np.median(rel[:, bi == x, 0, 1]):6.2f}" for x in range(8)))
    if NT == 100:       # the wide kernel also records where each wave ran (slot 13)
        raw = (a[0, :, 0, 13] / 0.01).round().astype(np.int64)
        xcc, hw = raw & 0xff, raw >> 8
        cu, sh, se, simd = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3
        fin_b = np.median(rel[:, :, 0, 11], axis=0)
        strm_b = np.median(rel[:, :, 1, 4], axis=0)
        print("block % 8 -> XCC_ID:", [sorted(set(xcc[bi == x].tolist())) for x in range(8)])
        # which SIMDs the traders / streamers of one CU sit on (two waves per SIMD are resident per CU)
        raw_s = (a[0, :, 1, 13] / 0.01).round().astype(np.int64)
        simd_s = ((raw_s >> 8) >> 4) & 3
        from collections import Counter
        cu_key = xcc * 1000 + se * 100 + sh * 50 + cu
        pat_t, pat_s, same = Counter(), Counter(), 0
        for key in set(cu_key.tolist()):
            m = cu_key == key
            pat_t[tuple(sorted(simd[m].tolist()))] += 1
            pat_s[tuple(sorted(simd_s[m].tolist()))] += 1
        same = int((simd == simd_s).sum())
        print("traders per CU, their SIMD ids (sorted) -> number of CUs:", dict(pat_t.most_common(8)))
        print("streamers per CU, their SIMD ids -> number of CUs:", dict(pat_s.most_common(8)))
        print(f"blocks whose trader and streamer share a SIMD: {same} of {nb}")
        # traders that share their SIMD with another trader vs those that do not
        shared = np.zeros(nb, bool)
        for key in set(cu_key.tolist()):
            idx = np.flatnonzero(cu_key == key)
            c = Counter(simd[idx].tolist())
            for i in idx:
                shared[i] = c[simd[i]] > 1
        if shared.any() and (~shared).any():
            for k in (5, 7, 9, 11):
                print(f"   trader stamp {k}: shares its SIMD with another trader {np.median(rel[:, shared, 0, k]):.2f} us "
                      f"({int(shared.sum())} blocks), does not {np.median(rel[:, ~shared, 0, k]):.2f} us")
        for name, key in (("XCC", xcc), ("SE", se), ("SH", sh), ("CU", cu), ("SIMD", simd)):
            ks = sorted(set(key.tolist()))
            print(f"   median trader finish / streamer done by {name}:",
                  " ".join(f"{k}:{np.median(fin_b[key == k]):.1f}/{np.median(strm_b[key == k]):.1f}" for k in ks))
    last = rel[:, :, 0, 11].max(axis=1)
    print(f"last trader finishes at {np.median(last):.2f} us; last streamer at "
          f"{np.median(rel[:, :, 1, 4].max(axis=1)):.2f} us")


if __name__ == "__main__":
    main()
