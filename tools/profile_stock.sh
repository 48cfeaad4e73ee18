#!/bin/bash
# rocprofv3 passes for the stock_step kernel (run on the GPU box via gpurun):
#   1. --kernel-trace --stats      -> per-kernel average duration
#   2. --pmc FETCH_SIZE            -> HBM read bytes  (own pass: TCC slots, MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE            -> HBM write bytes (own pass)
# Usage: bash tools/profile_stock.sh <tag> [extra bench.py args]   (outputs under gpurun_out/prof_<tag>/)
set -e
TAG=${1:-r01}
shift || true
EXTRA="$@"
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-cpu-baseline --prewarm 0 --steps 600 --warmup 100 $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 $ROOT/tools/summarize_profile.py $OUT | tee $OUT/summary.txt
