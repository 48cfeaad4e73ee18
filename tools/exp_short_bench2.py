#!/usr/bin/env python3
"""Prewarm recipe check: [2048 launches, sync, 64 launches, sync, reset, 5 warmup, sync] then K=20 timed;
repeated 5 times from a cold-ish state (sleep 0.5 s between)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from finrl_amd import StockPanel
from finrl_amd.vec_env import VecStockTradingEnv
E, N = 65536, 30
close, tech, risk = bench.synth_panel()
dev = torch.device("cuda", 0)
env = VecStockTradingEnv(StockPanel(close, tech, risk), E, device=dev, **bench.ENV_KW)
env.reset()
pool = [torch.rand(E, N, device=dev) * 2 - 1 for _ in range(16)]
def eager(k):
    for i in range(k):
        env.step(pool[i & 15])
def timed(K):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); eager(K); e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / K, e0.elapsed_time(e1) * 1e3 / K
for mode in ("tail64", "notail", "spin"):
    for rep in range(4):
        time.sleep(0.5)
        env.reset(); eager(2048); torch.cuda.synchronize()
        if mode == "tail64":
            eager(64); torch.cuda.synchronize()
        elif mode == "spin":
            t = time.perf_counter()
            while time.perf_counter() - t < 0.003: pass
        env.reset(); eager(5)
        w, d = timed(20)
        print(f"{mode} rep {rep}: wall {w:.2f} events {d:.2f} us/step")
