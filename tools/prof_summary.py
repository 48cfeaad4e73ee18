#!/usr/bin/env python3
"""Condense one tools/prof_one.sh output directory: kernel average / min / max and the per-dispatch
durations of the kernel whose name contains <needle> (timed dispatches only: the last `calls_timed`),
FETCH_SIZE / WRITE_SIZE medians per launch with the gfx950 correction of MI355X_MICROARCH.md (HBM
section): hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main():
    d, needle = sys.argv[1], sys.argv[2]
    res = dict(needle=needle, kernel={})
    f = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        res["top_kernels"] = [dict(name=r["Name"][:140], calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]),
                                   min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]), pct=float(r["Percentage"]))
                              for r in rows[:6]]
        for r in rows:
            if needle in r["Name"]:
                res["kernel"] = dict(name=r["Name"][:200], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                     min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3)
                break
    f = find(os.path.join(d, "trace"), "*kernel_trace.csv")
    if f:
        dur = []
        for r in csv.DictReader(open(f)):
            if needle in r.get("Kernel_Name", ""):
                dur.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        dur.sort()
        res["dispatch_us"] = [round(x[1], 2) for x in dur]
        if len(dur) > 1:        # gaps between consecutive dispatches (end -> next start is not in the csv: start -> start)
            st = [x[0] for x in dur]
            res["start_to_start_us"] = [round((b - a) / 1e3, 2) for a, b in zip(st[:-1], st[1:])]

    def med(sub, ctr):
        f = find(os.path.join(d, sub), "*counter_collection.csv")
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
             if needle in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr] if f else []
        return (sorted(v)[len(v) // 2], len(v)) if v else (None, 0)
    f_kb, nf = med("pmc_fetch", "FETCH_SIZE")
    w_kb, nw = med("pmc_write", "WRITE_SIZE")
    res.update(fetch_size_kib=f_kb, write_size_kib=w_kb, n_fetch_samples=nf, n_write_samples=nw)
    if f_kb is not None and w_kb is not None:
        res["hbm_bytes_per_launch"] = (2 * f_kb + w_kb) * 1024
        res["hbm_bytes_per_launch_raw"] = (f_kb + w_kb) * 1024
    print(json.dumps(res))


if __name__ == "__main__":
    main()
