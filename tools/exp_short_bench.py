#!/usr/bin/env python3
"""How much of a 20-step bench is pipeline fill?  Eager vs hipGraph launches, with / without an
untimed prewarm (clock ramp).  GPU box only."""
import os, sys, time, json
import numpy as np
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

def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6, e0.elapsed_time(e1) * 1e3

def eager(k):
    for i in range(k):
        env.step(pool[i & 15])

for label, prewarm in (("cold", 0), ("prewarm3000", 3000)):
    env.reset(); eager(prewarm); env.reset(); eager(5)
    for K in (20, 20, 20, 100, 1000):
        w, d = timed(lambda: eager(K))
        print(f"{label} eager K={K}: wall {w / K:.2f} us/step, events {d / K:.2f} us/step")
# graph
side = torch.cuda.Stream()
for K in (20, 100):
    g = torch.cuda.CUDAGraph()
    env.reset(); eager(5)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        eager(K)
    for rep in range(3):
        w, d = timed(g.replay)
        print(f"graph K={K}: wall {w / K:.2f} us/step, events {d / K:.2f} us/step")
