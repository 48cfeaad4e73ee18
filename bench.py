#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched StockTradingEnv hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU, RCCL).  Prints ONE JSON line on rank 0.

Workload = BASELINE.json configs[1]: 65,536 vectorised StockTradingEnv per GPU, DOW30 x 8
indicators, T = 2893 days (Stock_NeurIPS2018 split), uniform(-1,1) random actions, synthetic
panel.  A "step" is ONE launch of the step kernel over the whole batch
(E env-steps), including the terminal/auto-reset step of every episode.  Inputs (panel,
state, a pool of pre-generated action batches) are resident in HBM before the timed region.

N > 1: independent env shards per rank (weak scaling), no data-path collective; the only
collective is the RCCL all_gather of per-env episode returns at each episode end
(SURVEY.md 8e), inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

E_PER_GPU = 65_536
N_TICKERS, N_TECH, N_DAYS = 30, 8, 2893
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(N, K):
    """actions 4N + state read (12+4N) + state write (12+4N) + obs write
    4(1+2N+KN) + reward 4 + done 1  ==  4KN + 20N + 33  (1593 B at N=30, K=8)."""
    return 4 * K * N + 20 * N + 33


def synth_panel(T=N_DAYS, N=N_TICKERS, K=N_TECH, seed=0):
    """close = f32(100*exp(cumsum N(0, 0.01^2))), tech = f32(N(0,1)) with no
    exact 1.0 in indicator 0, risk = f32(|N(0,30)|); default_rng(0)."""
    rng = np.random.default_rng(seed)
    close = (100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))).astype(np.float32)
    tech = rng.normal(0, 1, (T, K, N)).astype(np.float32)
    tech[:, 0, :][tech[:, 0, :] == 1.0] = 0.5
    risk = np.abs(rng.normal(0, 30, T)).astype(np.float32)
    return close.astype(np.float64), tech.astype(np.float64), risk.astype(np.float64)


ENV_KW = dict(hmax=100, initial_amount=1_000_000, buy_cost_pct=1e-3, sell_cost_pct=1e-3,
              reward_scaling=1e-4)      # Stock_NeurIPS2018_SB3.py:251-272 (scalar costs)


def cpu_baseline(close, tech, risk, budget_s=12.0):
    """Time the oracle (oracle/stock_oracle.c, single thread) on a bounded sample of the same
    workload: the first `Ec` envs of the batch, same panel, same action distribution."""
    from oracle.stock import StockOracle
    Ec, chunk = 256, 200
    orc = StockOracle(close, tech, risk, n_envs=Ec, **ENV_KW)
    orc.reset()
    rng = np.random.default_rng(1234)
    acts = rng.uniform(-1, 1, (8, Ec, close.shape[1])).astype(np.float32)
    orc.vec_step(acts[0])               # warm
    steps, t0 = 0, time.perf_counter()
    while True:
        for j in range(chunk):
            orc.vec_step(acts[j & 7])
        steps += chunk
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return dict(value=Ec * steps / dt, unit="env-steps/s", cores=1, kind="port",
                sample=f"{Ec} envs x {steps} steps ({dt:.1f} s) of the same DOW30x8 workload, "
                       "oracle/stock_oracle.c single thread, obs included "
                       f"(host has {os.cpu_count()} cores; the reference itself -- Python/pandas, "
                       "one core of the build container -- runs this env at ~153 env-steps/s, "
                       "SURVEY.md section 6; it cannot travel to the GPU box)")


def parity_sample(close, tech, risk, dev, n_envs=256, n_steps=300):
    """Second half of BASELINE.json's metric ("reward max|delta| vs ref"), measured live inside
    the cpu_baseline leg: the oracle (pinned bit-exact to the reference) and the HIP env replay the
    same random action streams; reports the worst differences over all envs and steps."""
    import torch
    from finrl_amd import StockPanel
    from finrl_amd.vec_env import VecStockTradingEnv
    from oracle.stock import StockOracle
    Ts = min(close.shape[0], 120)              # two episode ends inside n_steps
    c, t, r = close[:Ts], tech[:Ts], risk[:Ts]
    orc = StockOracle(c, t, r, n_envs=n_envs, **ENV_KW)
    env = VecStockTradingEnv(StockPanel(c, t, r), n_envs, device=dev, **ENV_KW)
    ok_obs = bool(np.array_equal(env.reset().cpu().numpy(), orc.reset().astype(np.float32)))
    rng = np.random.default_rng(99)
    d_rew = d_asset = 0.0
    n_hold = n_done = n_tot = 0
    for s in range(n_steps):
        a = rng.uniform(-1, 1, (n_envs, close.shape[1])).astype(np.float32)
        g_obs, g_rew, g_done, _ = env.step(torch.from_numpy(a).to(dev))
        o_obs, o_rew, o_done, _ = orc.vec_step(a)
        d_rew = max(d_rew, float(np.max(np.abs(g_rew.cpu().numpy().astype(np.float64) - o_rew.astype(np.float32)))))
        st, os_ = env.state_numpy(), orc.state()
        ga = st["cash"] + (st["shares"] * c[st["day"]]).sum(1)
        oa = os_["cash"] + (os_["shares"] * c[os_["day"]]).sum(1)
        d_asset = max(d_asset, float(np.max(np.abs(ga - oa) / np.abs(oa))))
        n_hold += int((st["shares"] == os_["shares"]).all(1).sum())
        n_done += int((g_done.cpu().numpy().astype(bool) == o_done).sum())
        n_tot += n_envs
        ok_obs = ok_obs and bool(np.array_equal(g_obs.cpu().numpy(), o_obs.astype(np.float32)))
    return dict(envs=n_envs, steps=n_steps, reward_max_abs_delta=d_rew,
                total_asset_max_rel_delta=d_asset, holdings_match_rate=n_hold / n_tot,
                done_match_rate=n_done / n_tot, observations_identical=ok_obs)


def cpu_baseline_threads(close, tech, risk, n_threads, budget_s=6.0):
    """Same oracle, one independent env shard per thread (ctypes releases the GIL inside the C
    call): what the host cores this process may use deliver together.  Reported beside
    `cpu_baseline` (which stays the single-thread figure), never a target."""
    import threading
    from oracle.stock import StockOracle
    Ec, chunk = 256, 100
    rng = np.random.default_rng(4321)
    acts = rng.uniform(-1, 1, (8, Ec, close.shape[1])).astype(np.float32)
    orcs = [StockOracle(close, tech, risk, n_envs=Ec, **ENV_KW) for _ in range(n_threads)]
    for o in orcs:
        o.reset()
        o.vec_step(acts[0])
    counts = [0] * n_threads
    stop = time.perf_counter() + budget_s

    def work(k):
        o = orcs[k]
        while time.perf_counter() < stop:
            for j in range(chunk):
                o.vec_step(acts[j & 7])
            counts[k] += chunk

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return dict(value=Ec * sum(counts) / dt, unit="env-steps/s", cores=n_threads, kind="port",
                sample=f"{n_threads} threads x {Ec} envs, {sum(counts)} steps in total ({dt:.1f} s), "
                       "same workload and oracle as cpu_baseline")


def bench_portfolio(args, torch, dev):
    """BASELINE.json configs[2]: 65,536 vectorised StockPortfolioEnv, DOW30, K=8 (side metric;
    the driver's default line is the stock env)."""
    from finrl_amd.panel import PortfolioPanel
    from finrl_amd.vec_portfolio import VecStockPortfolioEnv
    E, N, K, T = args.envs_per_gpu, N_TICKERS, N_TECH, N_DAYS
    close, tech, _ = synth_panel()
    rets = np.diff(np.log(close), axis=0, prepend=np.log(close[:1]))
    cov = np.einsum("ti,tj->tij", rets, rets).astype(np.float32).astype(np.float64)
    env = VecStockPortfolioEnv(PortfolioPanel(close, cov, tech), E, device=dev)
    env.reset()
    pool = [torch.rand(E, N, device=dev) for _ in range(8)]
    for i in range(args.warmup):
        env.step(pool[i & 7])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        env.step(pool[i & 7])
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    B = 4 * N + 24 + 4 * N * (N + K) + 5           # 4709 B at N=30, K=8
    per = e0.elapsed_time(e1) * 1e-3 / args.steps
    ach = B * E / per / 1e9
    print(json.dumps({
        "metric": "env-steps/sec, vectorized StockPortfolioEnv (DOW30, 8 indicators)",
        "value": E * args.steps / wall, "unit": "env-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{E} vectorized StockPortfolioEnv, DOW30 x 8, T={T}",
                   "envs_per_gpu": E},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": side_traffic("portfolio", E),
                     "kernel": "portfolio_step_kernel", "bytes_per_env_step": B,
                     "avg_launch_us": per * 1e6}}), flush=True)


def side_traffic(kind, E):
    """PMC-measured HBM bytes per launch of a sibling kernel (profiles/side_traffic.json), or None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "side_traffic.json")))
        if tj.get("envs_per_gpu") == E:
            return tj["envs"][kind]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def bench_side(args, torch, dev, kind):
    """Side metrics: multi-crypto env (BASELINE configs[4] shape: 10 pairs x 4 indicators,
    1-minute bars) and the array-state (_np) stock env, single GPU."""
    rng = np.random.default_rng(0)
    E = args.envs_per_gpu
    if kind == "crypto":
        from finrl_amd.vec_crypto import VecCryptoEnv
        T, N, W = 43_200, 10, 40
        price = 10.0 ** rng.uniform(0, 4.5, N) * np.exp(
            np.cumsum(rng.normal(0, 0.0005, (T, N)), axis=0))
        env = VecCryptoEnv({"price_array": price, "tech_array": rng.normal(0, 3000, (T, W))}, E,
                           device=dev)
        B = 4 * N + 2 * (28 + 4 * N) + 4 * (1 + N + W) + 5          # 385
        name, kern = "vectorized CryptoEnv (10 pairs, 4 indicators/pair)", "crypto_kernel"
    elif kind in ("cashpenalty", "stoploss"):
        from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv, VecStopLossEnv
        T, N, Cc = N_DAYS, N_TICKERS, 5
        close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
        panel = CashPenaltyPanel(close, rng.normal(0, 10, (T, N, Cc)), np.abs(rng.normal(0, 30, T)))
        cls = VecCashPenaltyEnv if kind == "cashpenalty" else VecStopLossEnv
        env = cls(panel, E, hmax=2_000, random_start=True, device=dev)
        books = 1 if kind == "cashpenalty" else 6
        # actions + 2 x (cash f64, date/start i32, books f64[N]) + obs + reward/done
        B = 4 * N + 2 * (16 + 8 * books * N) + 4 * (1 + N + N * Cc) + 5
        name = f"vectorized {cls.env_name.split('-')[0]} (30 assets x 5 columns, random starts)"
        kern = f"{kind}_kernel"
    else:
        from finrl_amd.vec_stocknp import VecStockTradingEnvNP
        T, N, K = N_DAYS, N_TICKERS, N_TECH
        close, tech, risk = synth_panel()
        env = VecStockTradingEnvNP({"price_array": close, "tech_array": tech.transpose(0, 2, 1)
                                    .reshape(T, N * K), "turbulence_array": risk * 2,
                                    "if_train": False}, E, device=dev)
        B = 4 * N * K + 32 * N + 73                                  # 1993
        name, kern = "vectorized array-state StockTradingEnv (DOW30 x 8)", "stocknp_kernel"
    env.reset()
    pool = [torch.rand(E, N, device=dev) * 2 - 1 for _ in range(8)]
    for i in range(args.warmup):
        env.step(pool[i & 7])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        env.step(pool[i & 7])
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = e0.elapsed_time(e1) * 1e-3 / args.steps
    ach = B * E / per / 1e9
    print(json.dumps({
        "metric": f"env-steps/sec, {name}", "value": E * args.steps / wall,
        "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{E} {name}", "envs_per_gpu": E},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": side_traffic(kind, E), "kernel": kern,
                     "bytes_per_env_step": B, "avg_launch_us": per * 1e6}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="stock", choices=["stock", "portfolio", "crypto", "stocknp", "cashpenalty",
                                                       "stoploss"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3 * N_DAYS)
    ap.add_argument("--warmup", type=int, default=N_DAYS)
    ap.add_argument("--envs-per-gpu", type=int, default=E_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="disable on-device Sharpe stats")
    ap.add_argument("--desync", action="store_true",
                    help="per-env random start offsets (defeats panel-row broadcast)")
    ap.add_argument("--action-pool", type=int, default=16)
    ap.add_argument("--tickers", type=int, default=N_TICKERS,
                    help="30 = DOW30 (headline); 100 = NASDAQ-100 shape (BASELINE configs[3])")
    ap.add_argument("--turbulence-pct", type=float, default=None,
                    help="turbulence_threshold = this percentile of the synthetic risk series")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from finrl_amd import StockPanel
    from finrl_amd.distributed import gather_episode_returns
    from finrl_amd.vec_env import VecStockTradingEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    if args.env == "portfolio":
        assert world == 1, "portfolio side-bench is single-GPU"
        return bench_portfolio(args, torch, dev)
    if args.env in ("crypto", "stocknp", "cashpenalty", "stoploss"):
        assert world == 1, "side benches are single-GPU"
        return bench_side(args, torch, dev, args.env)

    E, N, K, T = args.envs_per_gpu, args.tickers, N_TECH, N_DAYS
    close, tech, risk = synth_panel(N=N)
    thr = None if args.turbulence_pct is None else float(np.percentile(risk, args.turbulence_pct))
    env = VecStockTradingEnv(StockPanel(close, tech, risk), E, device=dev,
                             track_stats=not args.no_stats, auto_reset=True,
                             turbulence_threshold=thr, **ENV_KW)
    env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand(E, N, generator=gen, device=dev) * 2 - 1 for _ in range(args.action_pool)]
    if args.desync:     # walk every env to a random day by masked resets at random steps
        offs = torch.randint(0, T - 1, (E,), generator=gen, device=dev)
        env.state["day"].copy_(offs.to(torch.int32))
        env.state["price_day"].copy_(offs.to(torch.int32))

    gathered = None
    step_in_ep = 0

    def run(n):
        nonlocal step_in_ep, gathered
        for i in range(n):
            env.step(pool[i % len(pool)])
            step_in_ep += 1
            if world > 1 and not args.desync and step_in_ep == T:
                # episode end on every rank: gather per-env episode returns (RCCL, 256 KB/rank)
                gathered = gather_episode_returns(env.episode_return(), world * E)
            if step_in_ep == T:
                step_in_ep = 0

    run(args.warmup)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    if rank == 0:
        B = algorithmic_bytes(N, K)
        per_launch_s = dev_ms * 1e-3 / args.steps        # HIP events on the launch stream
        achieved = B * E / per_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("envs_per_gpu") == E and tj.get("kernel") == "stock_step" and \
                        tj.get("tickers") == N and thr is None:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": ("env-steps/sec at N parallel envs (DOW30, 8 indicators)" if N == 30 else
                       f"env-steps/sec at N parallel envs ({N} tickers, 8 indicators)"),
            "value": world * E * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{E} vectorized StockTradingEnv per GPU, "
                                   + ("DOW30" if N == 30 else f"{N} tickers") +
                                   f" x 8 indicators, T={T}, random actions"
                                   + (f", turbulence threshold p{args.turbulence_pct:g}" if thr is not None else "")
                                   + (", desynchronised start days" if args.desync else ""),
                       "envs_per_gpu": E, "global_envs": world * E, "tickers": N,
                       "indicators": K, "days": T, "track_stats": not args.no_stats,
                       "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "stock_step_kernel", "bytes_per_env_step": B,
                         "avg_launch_us": per_launch_s * 1e6},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(close, tech, risk)
            out["cpu_baseline"]["parity_sample"] = parity_sample(close, tech, risk, dev)
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))    # the box's CPU share for one GPU
            if nthr > 1:
                out["cpu_baseline_all_cores"] = cpu_baseline_threads(close, tech, risk, nthr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
