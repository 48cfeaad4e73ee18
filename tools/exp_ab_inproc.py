#!/usr/bin/env python3
"""A/B of two builds of libfinenv.so inside ONE process on ONE env: same handle, same device buffers
(so the same physical placement), only the library that launches the step kernel alternates.
usage: python3 tools/exp_ab_inproc.py <variant .so> <n100|n30|portfolio|stocknp> [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FINENV_OBS_PLACEMENT", "first")
sys.path.insert(0, ROOT)


def main():
    variants, kind = sys.argv[1].split(","), sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import torch
    import bench
    from finrl_amd import _native as nat
    base_path = nat.LIB_PATH
    libs = {}
    envs = {}
    for name, path in [("base", base_path)] + [(os.path.basename(v), v) for v in variants]:
        setting = None
        if "@" in path:                      # "<lib.so>@VAR=value": same library, an environment setting per run
            path, setting = path.split("@", 1)
        nat._lib, nat.LIB_PATH = None, os.path.abspath(path)
        key = name if name == "base" else name[10:].replace(".so", "")
        libs[key] = nat.lib()
        envs[key] = setting
    nat._lib, nat.LIB_PATH = libs["base"], base_path
    dev = torch.device("cuda", 0)
    E = 65536
    if ":" in kind:                        # "<kind>:<envs>", e.g. n30:262144
        kind, e_txt = kind.split(":")
        E = int(e_txt)
    if kind.startswith("crypto") and kind != "crypto":      # crypto32768 / crypto65536 / crypto131072 / crypto262144
        E = int(kind[6:])
    a = dict(env="crypto", tickers=30, turbulence_pct=None) if kind.startswith("crypto") else \
        dict(env="portfolio", tickers=30, turbulence_pct=None) if kind == "portfolio" else \
        dict(env="stocknp", tickers=30, turbulence_pct=None) if kind == "stocknp" else \
        dict(env="stock", tickers=100 if kind == "n100" else 30, turbulence_pct=90.0 if kind == "n100" else None)
    args = type("A", (), dict(envs_per_gpu=E, action_pool=8, rollout=0, desync=False, no_stats=False, **a))()
    w = bench.build_workload(args, torch, dev, 0)
    env = w.env
    env.reset()
    for i in range(2000):
        env.step(w.pool[i % 8])
    torch.cuda.synchronize()
    for r in range(rounds):
        for name in libs:
            nat._lib = libs[name]
            for k2, v2 in envs.items():      # clear the other runs' settings, apply this one's
                if v2:
                    os.environ.pop(v2.split("=")[0], None)
            if envs.get(name):
                os.environ[envs[name].split("=")[0]] = envs[name].split("=")[1]
            if hasattr(env, "_step_args"):
                env._step_args = None            # cached function pointer of the previous library
            for i in range(100):
                env.step(w.pool[i % 8])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(500):
                env.step(w.pool[i % 8])
            e1.record()
            torch.cuda.synchronize()
            print(f"{kind} round {r} {name:8s} {e0.elapsed_time(e1) * 1e3 / 500:.2f} us", flush=True)


if __name__ == "__main__":
    main()
