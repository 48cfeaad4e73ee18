#!/bin/bash
# GPU check for one sibling env kernel: parity tests, bench line, phase timeline (diag build).
#   tools/r02_side.sh cashpenalty|stoploss|stocknp|crypto [tag]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
K=$1; TAG=${2:-s}
mkdir -p $ROOT/gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_gpu_${K}_parity.py tests/test_gpu_facades.py tests/test_gpu_harness.py -x -q > gpurun_out/r02/${TAG}_${K}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r02/${TAG}_${K}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 bench.py --env $K --no-cpu-baseline --steps 3000 --warmup 500 > gpurun_out/r02/${TAG}_${K}_bench.json 2> gpurun_out/r02/${TAG}_${K}_bench.err || exit 1
python3 -c "
import json,sys
j=json.loads(open('gpurun_out/r02/${TAG}_${K}_bench.json').read().strip().splitlines()[-1])
print('$K us/launch %.2f frac %.3f' % (j['roofline']['avg_launch_us'], j['roofline']['frac']))"
if [ -f finrl_amd/lib/libfinenv_diag.so ] && [ -f tools/phase_times_${K}.py ]; then
  timeout -k 10 200 python3 tools/phase_times_${K}.py 65536 > gpurun_out/r02/${TAG}_${K}_phase.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02/${TAG}_${K}_phase.txt
fi
