#!/usr/bin/env python3
"""Timing experiments for the stock_step kernel (GPU box only; not part of the product).

    python tools/sweep_stock.py --envs 16384,65536,262144 [--diag 0,1,2,...] [--steps 300]

With a diagnostic library (FINENV_LIB=finrl_amd/lib/libfinenv_diag.so) the FINENV_DIAG bits
skip phases: 1 = obs write, 2 = trade loops, 4 = sort, 8 = action tile load.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="65536")
    ap.add_argument("--diag", default="0")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-stats", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    close, tech, risk = bench.synth_panel()
    panel = StockPanel(close, tech, risk)
    for E in [int(x) for x in args.envs.split(",")]:
        env = VecStockTradingEnv(panel, E, track_stats=not args.no_stats, **bench.ENV_KW)
        env.reset()
        pool = [torch.rand(E, panel.N, device="cuda") * 2 - 1 for _ in range(8)]
        for d in [int(x) for x in args.diag.split(",")]:
            os.environ["FINENV_DIAG"] = str(d)
            for i in range(args.warmup):
                env.step(pool[i & 7])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(args.steps):
                env.step(pool[i & 7])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.steps
            print(json.dumps(dict(E=E, diag=d, us_per_step=round(us, 2),
                                  env_steps_per_s=round(E / us * 1e6),
                                  GBs=round(E * bench.algorithmic_bytes(30, 8) / us / 1e3, 1))),
                  flush=True)
        del env


if __name__ == "__main__":
    main()
