#!/bin/bash
# HBM traffic of the sibling step kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# separate passes (MI355X_MICROARCH.md, HBM section), one pair per env.
# Usage: bash tools/profile_side_pmc.sh <tag>  -> gpurun_out/prof_side_pmc_<tag>/<env>_traffic.json
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_side_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ENV in portfolio crypto stocknp cashpenalty stoploss; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CTR --output-format csv -d $OUT/${ENV}_$CTR -- python3 $ROOT/bench.py --env $ENV --steps 300 --warmup 50 --prewarm 0 --no-cpu-baseline > $OUT/${ENV}_$CTR.log 2>&1
  done
  python3 - "$OUT" "$ENV" <<'PY'
import csv, glob, json, os, sys
out, env = sys.argv[1], sys.argv[2]
needle = {"portfolio": "portfolio_step_kernel", "crypto": "crypto_kernel<false", "stocknp": "stocknp_kernel<false>",
          "cashpenalty": "cashpenalty_kernel<false", "stoploss": "stoploss_step"}[env]
def med(ctr):
    f = glob.glob(os.path.join(out, f"{env}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0]))
         if needle in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr] if f else []
    return sorted(v)[len(v) // 2] if v else None
f_kb, w_kb = med("FETCH_SIZE"), med("WRITE_SIZE")
res = dict(env=env, kernel=needle, fetch_size_kib=f_kb, write_size_kib=w_kb)
if f_kb is not None and w_kb is not None:
    res["hbm_bytes_per_launch"] = (2 * f_kb + w_kb) * 1024     # gfx950: FETCH_SIZE x2 (guide)
    res["hbm_bytes_per_launch_raw"] = (f_kb + w_kb) * 1024
json.dump(res, open(os.path.join(out, f"{env}_traffic.json"), "w"), indent=1)
print(json.dumps(res))
PY
done
