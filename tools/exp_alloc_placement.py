#!/usr/bin/env python3
"""Does the physical placement of the observation buffer decide the step time?  One process, one env:
the step kernel is timed (HIP events, 400 launches) writing into each of several freshly allocated
observation buffers (earlier ones kept alive, so every trial sits on other pages), then again into the
first ones (A/B/A).   usage: python3 tools/exp_alloc_placement.py [n100|n30|portfolio] [trials]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FINENV_OBS_PLACEMENT", "first")     # the raw allocation, no probe
sys.path.insert(0, ROOT)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "n100"
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import torch
    import bench
    dev = torch.device("cuda", 0)
    E = 65536
    if kind == "portfolio":
        args = type("A", (), dict(env="portfolio", envs_per_gpu=E, action_pool=8, rollout=0, tickers=30,
                                  turbulence_pct=None, desync=False, no_stats=False))()
    else:
        N = 100 if kind == "n100" else 30
        args = type("A", (), dict(env="stock", envs_per_gpu=E, action_pool=8, rollout=0, tickers=N,
                                  turbulence_pct=90.0 if N == 100 else None, desync=False, no_stats=False))()
    w = bench.build_workload(args, torch, dev, 0)
    env = w.env
    env.reset()
    rew, done = env.reward, env.done

    def timed(obs_view, n=400):
        for i in range(100):
            env.step(w.pool[i % 8], out=(obs_view, rew, done))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            env.step(w.pool[i % 8], out=(obs_view, rew, done))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for i in range(2000):                       # clock ramp
        env.step(w.pool[i % 8])
    torch.cuda.synchronize()
    D = env.obs.shape[1]
    P = env.obs.stride(0)
    print(f"{kind}: E={E} D={D} pitch={P}; own buffer: {timed(env.obs):.2f} us  (ptr {env.obs.data_ptr():#x})")
    bufs = []
    for t in range(trials):
        b = torch.empty(E, P, dtype=torch.float32, device=dev)
        bufs.append(b)
        print(f"trial {t}: ptr {b.data_ptr():#x}  {timed(b[:, :D]):.2f} us")
    for t in range(min(3, trials)):
        print(f"again {t}: {timed(bufs[t][:, :D]):.2f} us")
    print(f"own buffer again: {timed(env.obs):.2f} us")
    # the caching allocator hands out sub-ranges of 2 MiB-aligned segments; a direct hipMalloc for comparison
    torch.cuda.empty_cache()
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)      # 1 GiB block, carve aligned views out of it
    base = big.data_ptr()
    for off_mb in (0, 2, 64, 300, 600):
        off = off_mb << 20
        if off + E * P * 4 > big.numel():
            continue
        v = big[off:off + E * P * 4].view(torch.float32).view(E, P)
        print(f"1 GiB block + {off_mb} MiB: ptr {v.data_ptr():#x}  {timed(v[:, :D]):.2f} us")


if __name__ == "__main__":
    main()
