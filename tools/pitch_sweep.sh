#!/bin/bash
# observation row pitch sweep (FINENV_OBS_PITCH, floats) on one box: bench line per pitch, interleaved rounds
#   bash tools/pitch_sweep.sh "<bench args>" pitch...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
ARGS="$1"; shift
for round in 1 2 ${ROUNDS3:+3}; do
  for pt in "$@"; do
    FINENV_OBS_PITCH=$pt python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pitch $pt round $round us/launch %.2f frac %.3f'%(j['roofline']['avg_launch_us'], j['roofline']['frac']))"
  done
done
