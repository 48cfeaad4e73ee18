#!/bin/bash
# rocprofv3 --kernel-trace --stats for the sibling-env side benches (run on the GPU box via gpurun).
# Usage: bash tools/profile_side.sh <tag>   -> gpurun_out/prof_side_<tag>/<env>_kernel_stats.csv
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_side_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ENV in portfolio crypto stocknp cashpenalty stoploss; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$ENV -- python3 $ROOT/bench.py --env $ENV --steps 600 --warmup 100 --prewarm 0 --no-cpu-baseline > $OUT/$ENV.log 2>&1
  f=$(find $OUT/$ENV -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$OUT/${ENV}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows[:6]:
        w.writerow([r["Name"][:120], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
PY
  grep -h '"metric"' $OUT/$ENV.log | tail -1 > $OUT/${ENV}_bench.json
  head -4 $OUT/${ENV}_kernel_stats.csv
done
