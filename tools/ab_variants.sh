#!/bin/bash
# A/B of library variants on ONE box: N = 100 bench line per variant, three rounds, interleaved
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/${RDIR:-r03}
ARGS="${ARGS:---tickers 100 --turbulence-pct 90 --steps 2000 --warmup 500}"
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=$ROOT/finrl_amd/lib/libfinenv.so; else lib=$ROOT/finrl_amd/lib/variants/libfinenv_$v.so; fi
    FINENV_LIB=$lib python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $round us/launch %.2f frac %.3f'%(j['roofline']['avg_launch_us'], j['roofline']['frac']))"
  done
done
