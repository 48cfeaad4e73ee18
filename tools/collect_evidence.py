#!/usr/bin/env python3
"""Copy one `tools/evidence.sh <round> <tag>` pass (gpurun_out/, scratch) into the tracked files
under profiles/: per-kernel rocprofv3 stats, per-dispatch duration lists, PMC traffic, the
un-profiled bench lines, phase timelines, hbm_traffic.json / side_traffic.json (what bench.py reads
for roofline.traffic) and profiles/SUMMARY.md.     usage: python tools/collect_evidence.py r03 f"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def last_json(path):
    try:
        return json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        return None


def main():
    rd, tag = (sys.argv + ["r03", "f"])[1:3]
    G = os.path.join(ROOT, "gpurun_out", rd, f"ev_{tag}")
    sys.path.insert(0, ROOT)
    import bench
    profs = {}
    for name in ("stock_n30", "stock_n100", "stock_desync", "portfolio", "crypto_64k", "crypto_32k", "crypto_256k",
                 "stocknp", "cashpenalty", "stoploss", "drivercmd", "portfolio_long"):
        j = last_json(os.path.join(G, name, "summary.json"))
        if not j:
            continue
        profs[name] = j
        with open(os.path.join(P, f"{rd}_{name}_kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for r in j.get("top_kernels", []):
                w.writerow([r["name"], r["calls"], r["avg_ns"], r["min_ns"], r["max_ns"], r["pct"]])
        d = j.get("dispatch_us") or []
        if d:       # per-dispatch durations of the step kernel, in launch order (kernel trace)
            with open(os.path.join(P, f"{rd}_{name}_dispatch_us.txt"), "w") as fh:
                fh.write(f"# {j['kernel'].get('name', '')[:120]}\n# {len(d)} dispatches in launch order, microseconds "
                         "(rocprofv3 --kernel-trace; the first 100 are warm-up)\n")
                fh.write("\n".join(" ".join(f"{x:.2f}" for x in d[i:i + 20]) for i in range(0, len(d), 20)) + "\n")
        slim = {k: v for k, v in j.items() if k not in ("dispatch_us", "start_to_start_us", "top_kernels")}
        json.dump(slim, open(os.path.join(P, f"{rd}_{name}_pmc_summary.json"), "w"), indent=1)
    E = 65536
    # ---- what bench.py reads for roofline.traffic -----------------------------------------------
    recs = []
    for name, N, turb, desync in (("stock_n30", 30, False, False), ("stock_n100", 100, True, False),
                                  ("stock_desync", 30, False, True)):
        j = profs.get(name)
        if not j or not j.get("hbm_bytes_per_launch"):
            continue
        B = bench.algorithmic_bytes(N, 8) + (bench.panel_row_bytes(N, 8) if desync else 0)
        recs.append(dict(kernel=j["kernel"]["name"].split("::")[-1][:60], envs_per_gpu=E, tickers=N, indicators=8,
                         turbulence=turb, desync=desync, hbm_bytes_per_launch=j["hbm_bytes_per_launch"],
                         hbm_bytes_per_launch_raw=j["hbm_bytes_per_launch_raw"], fetch_size_kib=j["fetch_size_kib"],
                         write_size_kib=j["write_size_kib"], avg_kernel_us=j["kernel"]["avg_us"],
                         algorithmic_bytes=B * E, ratio=j["hbm_bytes_per_launch"] / (B * E),
                         source=f"profiles/{rd}_{name}_pmc_summary.json"))
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_one.sh), median over the "
            "launches of one run; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md "
            "(gfx950 FETCH_SIZE counts 128-B requests at 64 B)")
    if recs:
        json.dump(dict(note=note + "; the desynchronised record's algorithmic bytes include the per-env panel row "
                       "(B + 1084)", round=rd, records=recs), open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
    side = []
    for name, env, Ee in (("portfolio", "portfolio", E), ("crypto_64k", "crypto", E), ("crypto_32k", "crypto", 32768),
                          ("crypto_256k", "crypto", 262144), ("stocknp", "stocknp", E),
                          ("cashpenalty", "cashpenalty", E), ("stoploss", "stoploss", E)):
        j = profs.get(name)
        if not j or not j.get("hbm_bytes_per_launch"):
            continue
        b = last_json(os.path.join(G, name, "bench_under_trace.json"))
        Bv = b["roofline"]["bytes_per_env_step"] if b else None
        side.append(dict(env=env, envs_per_gpu=Ee, kernel=j["kernel"]["name"][:80],
                         hbm_bytes_per_launch=j["hbm_bytes_per_launch"], fetch_size_kib=j["fetch_size_kib"],
                         write_size_kib=j["write_size_kib"], avg_kernel_us=j["kernel"]["avg_us"],
                         algorithmic_bytes_per_launch=Bv * Ee if Bv else None,
                         ratio=j["hbm_bytes_per_launch"] / (Bv * Ee) if Bv else None,
                         bytes_formula=b["roofline"]["bytes_formula"] if b else None,
                         source=f"profiles/{rd}_{name}_pmc_summary.json"))
    if side:
        json.dump(dict(note=note + "; `ratio` divides by the algorithmic bytes bench.py uses (bytes_formula; for the "
                       "cash-penalty / stop-loss envs they include the per-env panel row that random starts force, "
                       "which L2 / Infinity Cache serve)", round=rd, records=side),
                  open(os.path.join(P, "side_traffic.json"), "w"), indent=1)
    # ---- bench lines ------------------------------------------------------------------------------
    names = ["driver", "stock", "desync", "n100", "portfolio", "crypto", "crypto32k", "crypto64k", "crypto256k",
             "crypto32k_rollout", "crypto32k_rollout_eager", "stocknp", "cashpenalty", "stoploss"]
    lines = {n: last_json(os.path.join(G, f"bench_{n}.json")) for n in names}
    with open(os.path.join(P, f"{rd}_bench_all.jsonl"), "w") as fh:
        for n in names:
            if lines[n]:
                fh.write(json.dumps(dict(line=n, **lines[n])) + "\n")
    for f in glob.glob(os.path.join(G, "phase_*.txt")):
        shutil.copy(f, os.path.join(P, f"{rd}_{os.path.basename(f)[6:-4]}_phase_timeline.txt"))
    for src, dst in (("riskpre.jsonl", f"{rd}_riskpre.jsonl"), ("default_bench.json", f"{rd}_default_bench.json"),
                     ("drivercmd_bench.json", f"{rd}_drivercmd_bench.json"),
                     ("portfolio_default.json", f"{rd}_portfolio_pool16.json"),
                     ("portfolio_--action-pool_1.json", f"{rd}_portfolio_pool1.json")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    # ---- SUMMARY.md ---------------------------------------------------------------------------------
    tr = {(r["tickers"], r["turbulence"], r["desync"]): r for r in recs}
    sd = {(r["env"], r["envs_per_gpu"]): r for r in side}
    rows = [("stock", "StockTradingEnv (headline, DOW30 x 8)", "stock_n30", tr.get((30, False, False))),
            ("driver", "same, driver command `--steps 20 --warmup 5` (prewarm 2,048)", "drivercmd", None),
            ("desync", "same, desynchronised start days (B + 1084)", "stock_desync", tr.get((30, False, True))),
            ("n100", "StockTradingEnv, 100 tickers x 8, turbulence p90 (configs[3] per-GPU slice)", "stock_n100",
             tr.get((100, True, False))),
            ("portfolio", "StockPortfolioEnv (252-day rolling covariance)", "portfolio", sd.get(("portfolio", E))),
            ("crypto32k", "CryptoEnv, 32,768 envs (configs[4] per-GPU slice)", "crypto_32k", sd.get(("crypto", 32768))),
            ("crypto64k", "CryptoEnv, 65,536 envs", "crypto_64k", sd.get(("crypto", E))),
            ("crypto256k", "CryptoEnv, 262,144 envs on one GPU", "crypto_256k", sd.get(("crypto", 262144))),
            ("crypto32k_rollout", "configs[4] slice + PPO rollout buffers + GAE, hipGraph per 16-step segment", None, None),
            ("crypto32k_rollout_eager", "same, eager launches", None, None),
            ("stocknp", "array-state StockTradingEnv", "stocknp", sd.get(("stocknp", E))),
            ("cashpenalty", "StockTradingEnvCashpenalty (random starts)", "cashpenalty", sd.get(("cashpenalty", E))),
            ("stoploss", "StockTradingEnvStopLoss (random starts)", "stoploss", sd.get(("stoploss", E)))]
    out = [f"# Bench summary, round {rd[1:].lstrip('0')} (E = 65,536 envs per GPU unless noted; one MI355X, ONE box for every line)", "",
           f"`us/step`, `env-steps/s`: un-profiled `bench.py` lines (`profiles/{rd}_bench_all.jsonl`; HIP events over the timed region).",
           f"`kernel us`: rocprofv3 `--kernel-trace --stats` average (min-max) of the step kernel in a separate profiled run of the same box",
           f"(`profiles/{rd}_*_kernel_stats.csv`, per-dispatch lists `{rd}_*_dispatch_us.txt`).  `HBM MB`: PMC FETCH_SIZE x 2 + WRITE_SIZE per launch",
           "(`hbm_traffic.json`, `side_traffic.json`).  `frac` = algorithmic bytes / step time / 8 TB/s; `hbm_frac` = counter bytes / step time / 8 TB/s.",
           "Plain dword stores top out at 6.0-6.2 TB/s on this chip (MI355X_MICROARCH.md): 0.75-0.78 is the ceiling of a write-dominated kernel.", "",
           "| workload | us/step | env-steps/s | B per env-step | frac | hbm_frac | kernel us (min-max) | HBM MB per launch (x algorithmic) |",
           "|---|---|---|---|---|---|---|---|"]
    for key, label, prof, trec in rows:
        j = lines.get(key)
        if not j:
            continue
        t_us = j["roofline"]["avg_launch_us"]
        pj = profs.get(prof) if prof else None
        k = pj["kernel"] if pj and pj.get("kernel") else None
        hb = trec["hbm_bytes_per_launch"] if trec else None
        ratio = trec.get("ratio") if trec else None
        out.append("| %s | %.2f | %.2e | %d | %.3f | %s | %s | %s |" % (
            label, t_us, j["value"], j["roofline"]["bytes_per_env_step"], j["roofline"]["frac"],
            "%.3f" % (hb / (t_us * 1e-6) / 8e12) if hb else "",
            "%.2f (%.1f-%.1f)" % (k["avg_us"], k["min_us"], k["max_us"]) if k else "",
            "%.1f (%.2fx)" % (hb / 1e6, ratio) if hb and ratio else ""))
    rp = [json.loads(x) for x in open(os.path.join(P, f"{rd}_riskpre.jsonl")) if x.strip()] \
        if os.path.exists(os.path.join(P, f"{rd}_riskpre.jsonl")) else []
    if rp:
        out += ["", f"Risk precompute (`tools/bench_riskpre.py`, `profiles/{rd}_riskpre.jsonl`):", ""]
        out += ["* " + json.dumps(r) for r in rp]
    d = last_json(os.path.join(P, f"{rd}_default_bench.json"))
    if d and "cpu_baseline" in d:
        out += ["", f"CPU baselines on the GPU box host (default `bench.py` run, `profiles/{rd}_default_bench.json`): oracle C single "
                "thread %.3g env-steps/s, %d threads %.3g, reference-shaped pandas/list Python env %.3g; live parity sample: %s." % (
                    d["cpu_baseline"]["value"], d.get("cpu_baseline_all_cores", {}).get("cores", 0),
                    d.get("cpu_baseline_all_cores", {}).get("value", 0), d.get("cpu_baseline_python", {}).get("value", 0),
                    json.dumps(d["cpu_baseline"].get("parity_sample", {}))[:300])]
    tl = sorted(os.path.basename(f) for f in glob.glob(os.path.join(P, f"{rd}_*_phase_timeline.txt")))
    out += ["", "In-kernel phase timelines (stamped diagnostic builds): " + ", ".join(f"`{t}`" for t in tl) + "."]
    open(os.path.join(P, "SUMMARY.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[8:]))


if __name__ == "__main__":
    main()
