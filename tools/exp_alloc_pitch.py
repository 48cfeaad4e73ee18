#!/usr/bin/env python3
"""Same process, same 1.5 GiB device block: the N = 100 step kernel writing observation rows of
different pitches / base offsets carved out of that one block (placement fixed, access pattern varied),
then the same pitches in freshly allocated buffers.  usage: python3 tools/exp_alloc_pitch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FINENV_OBS_PLACEMENT", "first")     # the raw allocation, no probe
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    E = 65536
    args = type("A", (), dict(env="stock", envs_per_gpu=E, action_pool=8, rollout=0, tickers=100,
                              turbulence_pct=90.0, desync=False, no_stats=False))()
    w = bench.build_workload(args, torch, dev, 0)
    env = w.env
    env.reset()
    rew, done = env.reward, env.done
    D = env.obs.shape[1]

    def timed(obs_view, n=300):
        for i in range(60):
            env.step(w.pool[i % 8], out=(obs_view, rew, done))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            env.step(w.pool[i % 8], out=(obs_view, rew, done))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for i in range(2000):
        env.step(w.pool[i % 8])
    torch.cuda.synchronize()
    big = torch.empty(3 << 29, dtype=torch.uint8, device=dev)

    def view(off_bytes, P):
        return big[off_bytes:off_bytes + E * P * 4].view(torch.float32).view(E, P)[:, :D]
    print(f"block ptr {big.data_ptr():#x}; own buffer {timed(env.obs):.2f} us")
    for P in (1001, 1008, 1016, 1024, 1032, 1040, 1056, 1072, 1088, 1104, 1120, 1152, 1280, 1536, 2048):
        print(f"pitch {P:5d} ({P * 4:5d} B, block stride {64 * P * 4:7d} B): {timed(view(0, P)):.2f} us")
    for off in (0, 4096, 65536, 1 << 20, (1 << 20) + 4096, 7 << 20, 256 << 20, (256 << 20) + 8192):
        print(f"pitch 1008 at +{off:>10d} B: {timed(view(off, 1008)):.2f} us")
    keep = []
    for P in (1008, 1056, 1088, 1152):
        for rep in range(3):
            b = torch.empty(E, P, dtype=torch.float32, device=dev)
            keep.append(b)
            print(f"fresh buffer pitch {P} #{rep} ptr {b.data_ptr():#x}: {timed(b[:, :D]):.2f} us")


if __name__ == "__main__":
    main()
