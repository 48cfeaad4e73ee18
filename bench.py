#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched StockTradingEnv hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU, RCCL).  Prints ONE JSON line on rank 0.  Started WITHOUT a torchrun environment and with
`--gpus N > 1`, it launches that very command itself as a child process (before anything touches
the GPU; never by exec) and exits with the child's code -- rank 0's JSON line arrives on the
inherited stdout.

Workload = BASELINE.json configs[1]: 65,536 vectorised StockTradingEnv per GPU, DOW30 x 8
indicators, T = 2893 days (Stock_NeurIPS2018 split), uniform(-1,1) random actions, synthetic
panel.  A "step" is ONE launch of the step kernel over the whole batch
(E env-steps), including the terminal/auto-reset step of every episode.  Inputs (panel,
state, a pool of pre-generated action batches) are resident in HBM before the timed region.

Timing: `--warmup W` untimed steps, then EXACTLY `--steps K` steps between barrier +
synchronize pairs (HIP events on the launch stream and the host clock; max over ranks).  Before
the warm-up an untimed, disclosed PREWARM phase (`prewarm_launches` in the JSON; default 2,048
launches when W < 1,024, then a reset) brings the GPU to its sustained clock: a cold 25-launch run
measures clock ramp, not the kernel (profiles/r02_short_bench.md: 22.7 us/step cold vs 21.2 after
the prewarm vs 20.9 in a 8,679-step run).

N > 1: independent env shards per rank (weak scaling), no data-path collective; the only
collective is the RCCL all_gather of per-env episode returns at each episode end,
inside the timed region -- and at least once per timed region (half way through and asynchronous
when no episode ends in it), so that a short scaling run still shows every rank taking part
(`rccl` in the JSON).

Other workloads (side metrics and BASELINE configs[2..4]): `--env portfolio|crypto|stocknp|riskpre|
cashpenalty|stoploss`, `--tickers 100 --turbulence-pct 90` (configs[3] per-GPU slice),
`--env crypto --envs-per-gpu 32768 --rollout 16` (configs[4] per-GPU slice: steps write straight
into [n_steps, E, .] rollout buffers, one GAE scan per segment, one hipGraph replay per segment).
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
PREWARM_DEFAULT = 2048


def algorithmic_bytes(N, K):
    """actions 4N + state read (12+4N) + state write (12+4N) + obs write
    4(1+2N+KN) + reward 4 + done 1  ==  4KN + 20N + 33  (1593 B at N=30, K=8)."""
    return 4 * K * N + 20 * N + 33


def panel_row_bytes(N, K):
    """one day's panel row, charged per env-step only in the desynchronised run
    (per-env rows defeat the broadcast): 4(K+1)N + 4 = 1084 B at N=30, K=8."""
    return 4 * (K + 1) * N + 4


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


# ------------------------------------------------------------------------------- CPU baseline
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


def cpu_baseline_python(close, tech, risk, budget_s=5.0):
    """a reference-SHAPED single env of this build's own authorship
    (oracle/pandas_env.py: DataFrame-backed, Python-list state, pandas `.loc[day]` per step), timed
    on one host core -- the apples-to-apples interpreter cost beside the C port."""
    from oracle.pandas_env import PandasStockEnv, make_frame
    Tb = min(close.shape[0], N_DAYS)
    env = PandasStockEnv(make_frame(close[:Tb], tech[:Tb], risk[:Tb]), **ENV_KW)
    env.reset()
    rng = np.random.default_rng(7)
    acts = rng.uniform(-1, 1, (64, close.shape[1])).astype(np.float32)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        _, _, done, _ = env.step(acts[steps & 63])
        steps += 1
        if done:
            env.reset()
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="env-steps/s", cores=1, kind="port",
                sample=f"1 env x {steps} steps ({dt:.1f} s), T={Tb}: pandas/list-state Python env "
                       "(oracle/pandas_env.py, validated against the reference fixtures)")


# ------------------------------------------------------------------------------- workloads
def side_traffic(kind, E):
    """PMC-measured HBM bytes per launch of a sibling kernel (profiles/side_traffic.json), or None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "side_traffic.json")))
        for rec in tj.get("records", []):              # one record per (env, batch size) profiled
            if rec.get("env") == kind and rec.get("envs_per_gpu") == E:
                return rec["hbm_bytes_per_launch"]
        if tj.get("envs_per_gpu") == E and kind in tj.get("envs", {}):
            return tj["envs"][kind]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def stock_traffic(E, N, thr, desync):
    """PMC-measured HBM bytes per launch of the stock step kernel (profiles/hbm_traffic.json)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        for rec in tj.get("records", [tj]):
            if rec.get("envs_per_gpu") == E and rec.get("tickers") == N and \
                    bool(rec.get("turbulence")) == (thr is not None) and \
                    bool(rec.get("desync")) == bool(desync):
                return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


class Workload:
    """What the timed loop needs: step(i), episode_return(), and the reporting metadata."""
    episode_len = None            # steps per episode when every env ends together, else None
    config_extra = {}

    def step(self, i):
        self.env.step(self.pool[i % len(self.pool)])

    def reset(self):
        self.env.reset()

    def episode_return(self):
        return self.env.episode_return()


def build_workload(args, torch, dev, rank):
    E = args.envs_per_gpu
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    rng = np.random.default_rng(0)          # panels are identical on every rank (replicated)
    w = Workload()
    w.kind = args.env
    lo = -1.0
    if args.env == "stock":
        from finrl_amd import StockPanel
        from finrl_amd.vec_env import VecStockTradingEnv
        N, K, T = args.tickers, N_TECH, N_DAYS
        close, tech, risk = synth_panel(N=N)
        thr = None if args.turbulence_pct is None else float(np.percentile(risk, args.turbulence_pct))
        w.env = VecStockTradingEnv(StockPanel(close, tech, risk), E, device=dev,
                                   track_stats=not args.no_stats, auto_reset=True,
                                   turbulence_threshold=thr, **ENV_KW)
        w.panel_arrays = (close, tech, risk)
        w.B = algorithmic_bytes(N, K) + (panel_row_bytes(N, K) if args.desync else 0)
        w.B_note = "4KN + 20N + 33" + (" + per-env panel row 4(K+1)N + 4" if args.desync else "")
        w.metric = ("env-steps/sec at N parallel envs (DOW30, 8 indicators)" if N == 30 else
                    f"env-steps/sec at N parallel envs ({N} tickers, 8 indicators)")
        w.workload = (f"{E} vectorized StockTradingEnv per GPU, " + ("DOW30" if N == 30 else f"{N} tickers")
                      + f" x 8 indicators, T={T}, random actions"
                      + (f", turbulence threshold p{args.turbulence_pct:g}" if thr is not None else "")
                      + (", desynchronised start days" if args.desync else ""))
        w.kernel = "stock_step_wide_kernel" if (N == 100 and ENV_KW["hmax"] <= 255) else "stock_step_kernel"
        w.traffic = stock_traffic(E, N, thr, args.desync)
        w.episode_len = None if args.desync else T
        w.config_extra = dict(tickers=N, indicators=K, days=T, track_stats=not args.no_stats,
                              obs_row_pitch_floats=int(w.env.obs.stride(0)))
        w.action_dim = N
        if args.desync:     # every env on its own day: defeats the panel-row broadcast
            def _desync():
                offs = torch.randint(0, T - 1, (E,), generator=gen, device=dev).to(torch.int32)
                w.env.state["day"].copy_(offs)
                w.env.state["price_day"].copy_(offs)
                w.env.state["start_day"].copy_(offs)
                w.env.refresh()         # price_day edited in place: re-evaluate the carried begin asset
                w.env.hint_desynchronised(True)
            w.after_reset = _desync
    elif args.env == "portfolio":
        from finrl_amd.panel import PortfolioPanel
        from finrl_amd.riskpre import rolling_covariance
        from finrl_amd.vec_portfolio import VecStockPortfolioEnv
        N, K, T = N_TICKERS, N_TECH, N_DAYS
        # cov[T, N, N] from a 252-day rolling window of the same kind of
        # synthetic returns, computed on the device by the risk-precompute kernel
        # (finenv_riskpre_rolling_cov); the series is generated 252 days longer and the env runs on
        # the last T days (the tutorial's frame likewise starts at the first full window)
        close, tech, _ = synth_panel(T=T + 252)
        cov = rolling_covariance(close, lookback=252, device=dev).cpu().numpy()
        close, tech = close[252:], tech[252:]
        cov = cov.astype(np.float32).astype(np.float64)
        w.env = VecStockPortfolioEnv(PortfolioPanel(close, cov, tech), E, device=dev)
        w.B = 4 * N + 24 + 4 * N * (N + K) + 5                       # 4709
        w.B_note = "4N + 24 + 4N(N+K) + 5"
        w.metric = "env-steps/sec, vectorized StockPortfolioEnv (DOW30, 8 indicators)"
        w.workload = f"{E} vectorized StockPortfolioEnv per GPU, DOW30 x 8, T={T}, 252-day rolling covariance"
        w.kernel, w.traffic, w.action_dim, lo = "portfolio_step_kernel", side_traffic("portfolio", E), N, 0.0
    elif args.env == "crypto":
        from finrl_amd.vec_crypto import VecCryptoEnv
        T, N, W = 43_200, 10, 40
        price = 10.0 ** rng.uniform(0, 4.5, N) * np.exp(
            np.cumsum(rng.normal(0, 0.0005, (T, N)), axis=0))
        w.env = VecCryptoEnv({"price_array": price, "tech_array": rng.normal(0, 3000, (T, W))}, E,
                             device=dev)
        w.B = 4 * N + 2 * (28 + 4 * N) + 4 * (1 + N + W) + 5          # 385
        w.B_note = "4N + 2(28+4N) + 4(1+N+40) + 5"
        w.metric = "env-steps/sec, vectorized CryptoEnv (10 pairs, 4 indicators/pair)"
        w.workload = f"{E} vectorized CryptoEnv per GPU, 10 pairs x 4 indicators, 1-minute bars, T={T}"
        w.kernel, w.traffic, w.action_dim = "crypto_kernel", side_traffic("crypto", E), N
    elif args.env in ("cashpenalty", "stoploss"):
        from finrl_amd.vec_cashpenalty import CashPenaltyPanel, VecCashPenaltyEnv, VecStopLossEnv
        T, N, Cc = N_DAYS, N_TICKERS, 5
        close = 50 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
        panel = CashPenaltyPanel(close, rng.normal(0, 10, (T, N, Cc)), np.abs(rng.normal(0, 30, T)))
        cls = VecCashPenaltyEnv if args.env == "cashpenalty" else VecStopLossEnv
        w.env = cls(panel, E, hmax=2_000, random_start=True, device=dev, seed=rank)
        books = 1 if args.env == "cashpenalty" else 6
        # actions + 2 x (cash f64, date/start i32, books f64[N]) + obs + reward/done, + the per-env
        # panel row (random starts: close f64[N] + info f32[N*C]) that no broadcast can serve
        w.B = 4 * N + 2 * (16 + 8 * books * N) + 4 * (1 + N + N * Cc) + 5 + (8 * N + 4 * N * Cc)
        w.B_note = "4N + 2(16 + 8*books*N) + 4(1+N+NC) + 5 + per-env panel row (8N + 4NC)"
        nm = cls.env_name.split("-")[0]
        w.metric = f"env-steps/sec, vectorized {nm} (30 assets x 5 columns, random starts)"
        w.workload = f"{E} vectorized {nm} per GPU, 30 assets x 5 columns, T={T}, random starts"
        w.kernel, w.traffic, w.action_dim = f"{args.env}_kernel", side_traffic(args.env, E), N
    elif args.env == "stocknp":
        from finrl_amd.vec_stocknp import VecStockTradingEnvNP
        T, N, K = N_DAYS, N_TICKERS, N_TECH
        close, tech, risk = synth_panel()
        w.env = VecStockTradingEnvNP({"price_array": close, "tech_array": tech.transpose(0, 2, 1)
                                      .reshape(T, N * K), "turbulence_array": risk * 2,
                                      "if_train": False}, E, device=dev)
        w.B = 4 * N * K + 32 * N + 73                                  # 1993
        w.B_note = "4NK + 32N + 73"
        w.metric = "env-steps/sec, vectorized array-state StockTradingEnv (DOW30 x 8)"
        w.workload = f"{E} vectorized array-state StockTradingEnv per GPU, DOW30 x 8, T={T}"
        w.kernel, w.traffic, w.action_dim = "stocknp_kernel", side_traffic("stocknp", E), N
        w.episode_len = T - 1
    else:
        raise ValueError(args.env)
    w.E = E
    A = w.action_dim
    w.pool = [torch.rand(E, A, generator=gen, device=dev) * (1.0 - lo) + lo
              for _ in range(args.action_pool)]
    if args.rollout:
        attach_rollout(w, args, torch, dev, gen)
    return w


def attach_rollout(w, args, torch, dev, gen):
    """BASELINE configs[4]: PPO rollout collection into device-resident [n_steps, E, .] buffers.
    Every step writes obs / reward / done straight into slice t (the C ABI takes output pointers:
    no staging copy), actions / values / log-probs of a stand-in policy (random tensors: no network is part of
    the env path) are stored by the env step's own launch where the env supports it (finenv_crypto_step_record:
    extra blocks beside the env blocks), else by one launch of their own (finenv_rollout_put), and every n_steps steps the GAE scan kernel runs
    (finenv_gae_scan).  Still one env step per `step`."""
    from finrl_amd.rollout import RolloutBuffer
    n = int(args.rollout)
    env = w.env
    D = env.obs.shape[1]
    buf = RolloutBuffer(n, w.E, D, w.action_dim, device=dev)
    vals = [torch.rand(w.E, generator=gen, device=dev) for _ in range(4)]
    lps = [-torch.rand(w.E, generator=gen, device=dev) for _ in range(4)]
    w.buf = buf
    w.config_extra = dict(w.config_extra, rollout_n_steps=n,
                          rollout_bytes=int(sum(t.numel() * t.element_size() for t in
                                                (buf.obs, buf.actions, buf.rewards, buf.dones,
                                                 buf.values, buf.log_probs, buf.advantages,
                                                 buf.returns))))
    w.workload += f" + PPO rollout collection ({n}-step segments, GAE scan per segment)"
    # per env-step on top of the env's own bytes: action copy (r+w), value / log-prob (w), and per
    # segment step the GAE scan's reward / value / done reads and advantage / return writes
    w.B += 8 * w.action_dim + 8 + (4 + 4 + 1 + 4 + 4)
    w.B_note += " + rollout: 8A + 8 + 17"
    state = dict(t=0)

    def one(t, i):
        buf.step(env, t, w.pool[i % len(w.pool)], vals[i & 3], lps[i & 3])

    def finish(i):
        buf.compute_returns_and_advantage(vals[i & 3], gamma=0.99, gae_lambda=0.95)
        buf.obs[0].copy_(buf.obs[n])

    if args.no_graph:
        def step(i):
            t = state["t"]
            one(t, i)
            t += 1
            if t == n:
                finish(i)
                t = 0
            state["t"] = t
        w.config_extra["launch"] = "eager"
    else:
        # a whole segment -- n x (env step that also records the policy's outputs) + GAE scan + carry-over copy -- is ONE
        # hipGraph: replayed with one host call per n steps (every step() of the envs is a plain
        # launch on the caller's stream: no allocation, sync or host read-back in the C ABI)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                for t in range(n):
                    one(t, t)
                finish(n - 1)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for t in range(n):
                one(t, t)
            finish(n - 1)
        w.graph = graph

        def step(i):
            state["t"] += 1
            if state["t"] == n:
                graph.replay()
                state["t"] = 0
        for name in ("steps", "warmup"):
            if getattr(args, name) % n:
                raise SystemExit(f"--rollout {n} with graph replay needs --{name} to be a multiple of {n}")
        w.config_extra["launch"] = f"hipGraph, one replay per {n}-step segment"

    def reset():
        buf.obs[0].copy_(env.reset())
        state["t"] = 0

    w.step, w.reset = step, reset


# ------------------------------------------------------------------------------- timed region
def timed_region(work, steps, warmup, prewarm, world, dist, sync, make_events, global_envs=None):
    """prewarm (untimed, then reset) -> warmup (untimed) -> EXACTLY `steps` steps between barrier +
    synchronize pairs.  Episode ends inside the timed region gather the per-env episode returns
    over all ranks; when no episode ends there (short runs) ONE asynchronous gather runs half way
    through and is waited for before the region closes, so every multi-rank run exercises the
    collective.  Returns wall seconds (max over
    ranks is taken by the caller), device milliseconds and the rccl record."""
    from finrl_amd.distributed import gather_episode_returns
    state = dict(in_ep=0, gathers=0, gathered=0)
    L = work.episode_len

    pending = []

    def gather(async_op=False):
        out = gather_episode_returns(work.episode_return(), global_envs, async_op=async_op)
        state["gathers"] += 1
        if async_op:
            pending.append(out)
        else:
            state["gathered"] = int(out.numel())

    def run(n, timed):
        # a timed region too short to contain an episode end gathers ONCE, half way through and
        # asynchronously: the collective runs on RCCL's stream beside the following steps and is
        # waited for before the region closes (at its end it would sit fully exposed on a 20-step
        # region: ~100 us of host + device time against 420 us of steps)
        forced_at = n // 2 if (timed and world > 1 and (L is None or state["in_ep"] + n < L)) else -1
        for i in range(n):
            work.step(i)
            if i == forced_at:
                gather(async_op=True)
            if L is not None:
                state["in_ep"] += 1
                if state["in_ep"] == L:
                    state["in_ep"] = 0
                    if world > 1 and timed:
                        gather()

    if prewarm > 0:                       # clock ramp; not part of any reported number
        run(prewarm, False)
        sync()
        run(64, False)                    # short tail: the host comes back from a long blocking wait
        sync()
        work.reset()
        if getattr(work, "after_reset", None):
            work.after_reset()
        state["in_ep"] = 0
    run(warmup, False)
    if world > 1:                         # untimed: RCCL sets up its channels at the first call
        gather_episode_returns(work.episode_return(), global_envs)
    ev0, ev1 = make_events()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    ev0.record()
    run(steps, True)
    for h in pending:
        state["gathered"] = int(h.wait().numel())
    ev1.record()
    sync()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    rccl = None
    if world > 1:
        rccl = dict(world=world, backend=dist.get_backend(), gathers=state["gathers"],
                    gathered=state["gathered"])
    return wall, ev0.elapsed_time(ev1), rccl


def riskpre_lines(cpu_baseline=True):
    """Side measurement: the risk precompute -- turbulence index + cov_list -- at the
    reference's panel sizes, one JSON line per shape; its cpu_baseline leg times the NumPy oracle on a
    bounded sample of days (the only place outside tests/ and smoke() that runs anything under
    oracle/, like the env benches' cpu_baseline)."""
    import time
    import torch
    from finrl_amd import riskpre
    rng = np.random.default_rng(0)
    for T, N in ((2893, 30), (2893, 100)):
        close = 100 * np.exp(np.cumsum(rng.normal(0, 0.01, (T, N)), axis=0))
        ct = torch.from_numpy(close).cuda()
        for _ in range(2):
            riskpre.calculate_turbulence(ct)
            riskpre.rolling_covariance(ct)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        riskpre.calculate_turbulence(ct)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        riskpre.rolling_covariance(ct)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        line = {"shape": [T, N], "turbulence_ms": (t1 - t0) * 1e3, "cov_list_ms": (t2 - t1) * 1e3,
                "turbulence_days_per_s": (T - 252) / (t1 - t0)}
        if cpu_baseline:
            from oracle import riskpre as orc
            c0 = time.perf_counter()
            orc.calculate_turbulence(close[:252 + 200])
            c1 = time.perf_counter()
            line["numpy_oracle_days_per_s"] = 200 / (c1 - c0)
        print(json.dumps(line), flush=True)


class _SelfTestWork(Workload):
    """`--env launcher-selftest`: no kernel, no GPU -- a counter that follows the Workload protocol
    so that the launch / rendezvous / gather / JSON plumbing of the N > 1 path can be run where no
    GPU exists (gloo on the CPU; tests/test_distributed_cpu.py).  Its `value` measures nothing."""

    def __init__(self, torch, E, rank, episode_len):
        self.torch, self.E, self.rank, self.episode_len = torch, E, rank, episode_len
        self.n = 0
        self.kind = "launcher-selftest"

    def step(self, i):
        self.n += 1

    def reset(self):
        pass

    def episode_return(self):
        return self.torch.arange(self.E, dtype=self.torch.float32) + self.rank * self.E


class _HostEvent:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def self_launch(n_ranks, argv):
    """`bench.py --gpus N` (N > 1) outside torchrun: run the driver's own command shape as a CHILD
    process -- one rank per GPU, rendezvous on 127.0.0.1 -- and return its exit code.  The parent
    has not imported torch, let alone touched the GPU; stdout / stderr are inherited, so rank 0's
    JSON line is this process's output."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what the host driver supports
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="stock", choices=["stock", "portfolio", "crypto", "stocknp", "cashpenalty",
                                                       "stoploss", "riskpre", "launcher-selftest"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3 * N_DAYS)
    ap.add_argument("--warmup", type=int, default=N_DAYS)
    ap.add_argument("--prewarm", type=int, default=None,
                    help="untimed launches before --warmup (clock ramp), followed by a reset; "
                         f"default {PREWARM_DEFAULT} when --warmup < 1024, else 0")
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
    ap.add_argument("--no-graph", action="store_true",
                    help="rollout mode: launch every copy / step eagerly instead of one hipGraph "
                         "replay per segment")
    ap.add_argument("--rollout", type=int, default=0,
                    help="collect into [n_steps, E, .] rollout buffers + GAE scan per segment "
                         "(BASELINE configs[4])")
    args = ap.parse_args(argv)
    if args.env == "riskpre":
        riskpre_lines(cpu_baseline=not args.no_cpu_baseline)
        return 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args.gpus, argv)        # before torch is imported: nothing has touched the GPU

    import torch
    import torch.distributed as dist

    selftest = args.env == "launcher-selftest"
    # FINENV_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo for the gather -- the N > 1 path with the real
    # kernels on a one-GPU box (RCCL needs one GPU per rank); tests/test_gpu_bench_contract.py.  Never a result.
    rehearsal = os.environ.get("FINENV_BENCH_REHEARSAL") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start N ranks with torchrun, or "
                         "run `bench.py --gpus N` without a torchrun environment (it launches them)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest or rehearsal:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if selftest else torch.device("cuda", 0 if rehearsal else local_rank)
    prewarm = args.prewarm if args.prewarm is not None else \
        (PREWARM_DEFAULT if args.warmup < 1024 else 0)

    if selftest:
        work = _SelfTestWork(torch, min(args.envs_per_gpu, 1024), rank, episode_len=None)
        prewarm = 0
    else:
        work = build_workload(args, torch, dev, rank)
    work.reset()
    if getattr(work, "after_reset", None):
        work.after_reset()

    def make_events():
        if selftest:
            return _HostEvent(), _HostEvent()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    wall, dev_ms, rccl = timed_region(work, args.steps, args.warmup, prewarm, world, dist,
                                      (lambda: None) if selftest else torch.cuda.synchronize,
                                      make_events, global_envs=world * work.E)
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    if rank == 0 and selftest:
        print(json.dumps({"metric": "launcher self-test (no kernel runs; value is meaningless)",
                          "value": world * work.E * args.steps / wall, "unit": "stub-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": wall * 1e3 / args.steps, "data": "stub",
                          "config": {"workload": "launcher-selftest", "envs_per_gpu": work.E,
                                     "global_envs": world * work.E},
                          "rccl": rccl, "stub_steps_run": work.n}), flush=True)
    elif rank == 0:
        E = work.E
        per_launch_s = dev_ms * 1e-3 / args.steps        # HIP events on the launch stream
        achieved = work.B * E / per_launch_s / 1e9
        out = {
            "metric": work.metric,
            "value": world * E * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "prewarm_launches": prewarm,
            "config": dict({"workload": work.workload, "envs_per_gpu": E,
                            "global_envs": world * E, "parallelism": f"env-shard x{world}"},
                           **work.config_extra),
            # frac: ALGORITHMIC bytes (bytes_formula x envs) / HIP-event launch time / 8 TB/s.
            # hbm_frac: the bytes the HBM counters saw for this kernel and shape (traffic: rocprofv3
            # PMC passes committed under profiles/, null when that shape was not profiled) over the
            # same time -- below frac where the formula charges rows that L2 / Infinity Cache serve,
            # above it where bookkeeping state and partial-segment writes add traffic.
            # wall_frac: frac with the host clock's ms_per_step instead of the HIP events.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": work.traffic,
                         "hbm_frac": (work.traffic / per_launch_s / 1e9 / HBM_PEAK_GBS
                                      if work.traffic else None),
                         "wall_frac": work.B * E / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "kernel": work.kernel, "bytes_per_env_step": work.B,
                         "bytes_formula": work.B_note, "avg_launch_us": per_launch_s * 1e6},
        }
        if rccl is not None:
            out["rccl"] = rccl
        if world == 1 and not args.no_cpu_baseline and args.env == "stock" and \
                args.tickers == N_TICKERS and not args.desync:
            close, tech, risk = work.panel_arrays
            out["cpu_baseline"] = cpu_baseline(close, tech, risk)
            out["cpu_baseline"]["parity_sample"] = parity_sample(close, tech, risk, dev)
            nthr = max(1, min(16, len(os.sched_getaffinity(0))))    # the box's CPU share for one GPU
            if nthr > 1:
                out["cpu_baseline_all_cores"] = cpu_baseline_threads(close, tech, risk, nthr)
            try:
                out["cpu_baseline_python"] = cpu_baseline_python(close, tech, risk)
            except ImportError:
                pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
