#!/usr/bin/env python3
"""Step time against the batch size (blocks per CU) for one env kind, one process per size.
usage: python3 tools/exp_esweep.py n100 40960 49152 57344 65536 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FINENV_OBS_PLACEMENT", "first")
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1]
    import torch
    import bench
    dev = torch.device("cuda", 0)
    for E in map(int, sys.argv[2:]):
        a = dict(env="portfolio", tickers=30, turbulence_pct=None) if kind == "portfolio" else \
            dict(env="stocknp", tickers=30, turbulence_pct=None) if kind == "stocknp" else \
            dict(env="stock", tickers=100 if kind == "n100" else 30, turbulence_pct=90.0 if kind == "n100" else None)
        args = type("A", (), dict(envs_per_gpu=E, action_pool=8, rollout=0, desync=False, no_stats=False, **a))()
        w = bench.build_workload(args, torch, dev, 0)
        env = w.env
        env.reset()
        for i in range(1500):
            env.step(w.pool[i % 8])
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(400):
                env.step(w.pool[i % 8])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 400)
        t = sorted(ts)[1]
        print(f"{kind} E={E:7d} blocks={(E + 63) // 64:5d}: {t:7.2f} us  {t * 1e3 / E:.3f} ns/env  "
              f"frac {w.B * E / (t * 1e-6) / 8e12:.3f}", flush=True)
        del w, env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
