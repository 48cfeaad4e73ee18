"""Times the risk precompute (turbulence index + cov_list) at the reference's panel sizes.
python tools/bench_riskpre.py  -> one JSON line per shape (GPU), plus the NumPy oracle timed on a
bounded sample of days for comparison."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from finrl_amd import riskpre
    from oracle import riskpre as orc
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
        sample = close[:252 + 200]
        c0 = time.perf_counter()
        orc.calculate_turbulence(sample)
        c1 = time.perf_counter()
        print(json.dumps({"shape": [T, N], "turbulence_ms": (t1 - t0) * 1e3,
                          "cov_list_ms": (t2 - t1) * 1e3,
                          "turbulence_days_per_s": (T - 252) / (t1 - t0),
                          "numpy_oracle_days_per_s": 200 / (c1 - c0)}), flush=True)


if __name__ == "__main__":
    main()
