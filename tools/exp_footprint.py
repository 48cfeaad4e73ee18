#!/usr/bin/env python3
"""Same process, same observation buffer: step time against the number of distinct action batches
cycled through (the bytes that pass through L2 / Infinity Cache between two visits of a line), and
against the batch size.   usage: python3 tools/exp_footprint.py [n100|n30|portfolio]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FINENV_OBS_PLACEMENT", "first")
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "n100"
    import torch
    import bench
    dev = torch.device("cuda", 0)
    for E in (65536, 49152, 32768):
        if kind == "portfolio":
            a = dict(env="portfolio", tickers=30, turbulence_pct=None)
        else:
            a = dict(env="stock", tickers=100 if kind == "n100" else 30,
                     turbulence_pct=90.0 if kind == "n100" else None)
        args = type("A", (), dict(envs_per_gpu=E, action_pool=16, rollout=0, desync=False, no_stats=False, **a))()
        w = bench.build_workload(args, torch, dev, 0)
        env = w.env
        env.reset()
        for i in range(1500):
            env.step(w.pool[i % 16])
        torch.cuda.synchronize()
        res = []
        for pool in (1, 2, 4, 8, 16, 1, 16):
            for i in range(100):
                env.step(w.pool[i % pool])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(400):
                env.step(w.pool[i % pool])
            e1.record()
            torch.cuda.synchronize()
            res.append((pool, e0.elapsed_time(e1) * 1e3 / 400))
        obs_mb = env.obs.shape[0] * env.obs.stride(0) * 4 / 1e6
        act_mb = w.pool[0].numel() * 4 / 1e6
        print(f"{kind} E={E} obs {obs_mb:.0f} MB, one action batch {act_mb:.1f} MB: " +
              "  ".join(f"pool{p}: {t:.2f} us ({t * 1e3 / E:.3f} ns/env)" for p, t in res))
        del w, env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
