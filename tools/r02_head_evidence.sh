#!/bin/bash
# Round-2 HEAD evidence (VERDICT r01 items 2,3,6): run on the GPU box via gpurun.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/r02
# exact driver command under rocprofv3 kernel trace, then un-profiled
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02/driver_trace -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/r02/driver_trace.log 2>&1 )
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/driver_cmd_1.json 2> gpurun_out/r02/driver_cmd_1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/driver_cmd_2.json 2> gpurun_out/r02/driver_cmd_2.err
python3 bench.py --gpus 1 --steps 20 --warmup 2893 --no-cpu-baseline > gpurun_out/r02/driver_cmd_longwarm.json 2> gpurun_out/r02/driver_cmd_longwarm.err
bash tools/profile_stock.sh r02_n100 --tickers 100 --turbulence-pct 90
python3 bench.py --no-cpu-baseline --tickers 100 --turbulence-pct 90 --steps 2000 --warmup 500 > gpurun_out/r02/bench_n100.json
python3 tools/phase_times.py 65536 100 > gpurun_out/r02/phase_n100.txt 2>&1
bash tools/profile_stock.sh r02_desync --desync
python3 bench.py --no-cpu-baseline --desync --steps 2000 --warmup 500 > gpurun_out/r02/bench_desync.json
echo done
