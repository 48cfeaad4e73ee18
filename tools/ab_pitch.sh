#!/bin/bash
# A/B of the observation row pitch on ONE box: packed [E][D] rows vs rows on 64-byte boundaries
ROOT=${GRAFT_REPO_ROOT:-$PWD}
ARGS="${ARGS:---steps 6000 --warmup 2000}"
for round in 1 2 3; do
  for v in packed aligned; do
    FINENV_OBS_PITCH=$v python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $round us/launch %.2f frac %.3f'%(j['roofline']['avg_launch_us'], j['roofline']['frac']))"
  done
done
