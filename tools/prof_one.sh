#!/bin/bash
# One workload under rocprofv3, three separate passes (MI355X_MICROARCH.md, HBM section: the two TCC
# counters do not fit one pass, and counters never share a run with the kernel trace):
#   1. --kernel-trace --stats   -> per-kernel average + the per-dispatch duration list
#   2. --pmc FETCH_SIZE         3. --pmc WRITE_SIZE
# usage: bash tools/prof_one.sh <outdir> <kernel-name needle> <bench.py args...>
set -e
OUT=$1; NEEDLE=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-cpu-baseline --prewarm 0 --steps ${PROF_STEPS:-600} --warmup 100 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
if [ -z "$NO_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
fi
python3 $ROOT/tools/prof_summary.py $OUT "$NEEDLE" > $OUT/summary.json
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_under_trace.json || true
cat $OUT/summary.json | python3 -c "import json,sys; j=json.load(sys.stdin); print('$NEEDLE', 'avg_us', j['kernel'].get('avg_us'), 'hbm_MB', (j.get('hbm_bytes_per_launch') or 0)/1e6)"
