#!/usr/bin/env python3
"""profiles/SUMMARY.md: one table over the latest bench lines and rocprofv3 kernel averages kept
under profiles/ (usage: python tools/summarize_benches.py r01_i)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def kernel_avg(path, needle):
    if not os.path.exists(path):
        return None
    for row in csv.DictReader(open(path)):
        if needle in row["Name"]:
            return float(row["AverageNs"]) / 1e3
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_i"
    rows = []
    b = json.load(open(os.path.join(P, f"{tag}_bench.json")))
    rows.append(("StockTradingEnv (headline, DOW30 x 8)", b, kernel_avg(
        os.path.join(P, f"{tag}_kernel_stats.csv"), "stock_step_kernel")))
    for env, needle in (("portfolio", "portfolio_step_kernel"), ("crypto", "crypto_kernel<false>"),
                        ("stocknp", "stocknp_kernel<false>"), ("cashpenalty", "cashpenalty_kernel<false"),
                        ("stoploss", "stoploss_kernel<false")):
        f = os.path.join(P, f"{tag}_{env}_bench.json")
        if os.path.exists(f):
            rows.append((env, json.loads(open(f).read().strip().splitlines()[-1]), kernel_avg(
                os.path.join(P, f"{tag}_{env}_kernel_stats.csv"), needle)))
    out = [f"# Bench summary ({tag}; E = 65,536 envs on one MI355X)", "",
           "`us/step` and `env-steps/s` are the bench's wall clock: un-profiled for the headline row, under",
           "rocprofv3 for the side envs (a few % slower; the cash-penalty / stop-loss steps also include the",
           "per-step draw of random start offsets, a ~5 us torch kernel).  `kernel us` is rocprofv3's",
           "average duration of the step kernel.  `frac` = algorithmic bytes / step time / 8 TB/s.", "",
           "| env | us/step | env-steps/s | B per env-step | frac of 8 TB/s | kernel us (rocprofv3) |",
           "|---|---|---|---|---|---|"]
    for name, b, k in rows:
        r = b["roofline"]
        out.append(f"| {name} | {b['ms_per_step'] * 1e3:.2f} | {b['value']:.3g} | "
                   f"{r['bytes_per_env_step']} | {r['frac']:.3f} | {k:.2f} |" if k else
                   f"| {name} | {b['ms_per_step'] * 1e3:.2f} | {b['value']:.3g} | "
                   f"{r['bytes_per_env_step']} | {r['frac']:.3f} | - |")
    rp = os.path.join(P, f"{tag}_riskpre.json")
    if os.path.exists(rp):
        out += ["", "Risk precompute (`tools/bench_riskpre.py`):", ""]
        for line in open(rp):
            d = json.loads(line)
            out.append(f"* T x N = {d['shape'][0]} x {d['shape'][1]}: turbulence index "
                       f"{d['turbulence_ms']:.2f} ms, cov_list {d['cov_list_ms']:.2f} ms "
                       f"({d['turbulence_days_per_s'] / d['numpy_oracle_days_per_s']:.0f}x the NumPy "
                       f"oracle on one host core)")
    open(os.path.join(P, "SUMMARY.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
