#!/bin/bash
# Round-2 evidence on ONE box: rocprofv3 kernel stats + PMC traffic for every step kernel, then
# the un-profiled bench lines.  Outputs under gpurun_out/ (copied into profiles/ afterwards).
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/r02
bash tools/profile_stock.sh r02f_n30 > /dev/null 2>&1
bash tools/profile_stock.sh r02f_n100 --tickers 100 --turbulence-pct 90 > /dev/null 2>&1
bash tools/profile_stock.sh r02f_desync --desync > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02/driver_trace_f -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/r02/driver_trace_f.log 2>&1 )
bash tools/profile_side.sh r02f > /dev/null 2>&1
bash tools/profile_side_pmc.sh r02f > /dev/null 2>&1
bash tools/r02_bench_all.sh f
python3 tools/bench_riskpre.py 2>/dev/null | grep shape > $ROOT/gpurun_out/r02/f_riskpre.jsonl
python3 tools/phase_times.py 65536 30 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_n30.txt
python3 tools/phase_times.py 65536 100 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_n100.txt
python3 tools/phase_times_crypto.py 32768 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_crypto.txt
python3 tools/phase_times_stocknp.py 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_stocknp.txt
python3 tools/phase_times_cashpenalty.py 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_cashpenalty.txt
python3 tools/phase_times_stoploss.py 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_stoploss.txt
python3 tools/phase_times.py 65536 30 desync 2>&1 | grep -v amdgpu.ids > $ROOT/gpurun_out/r02/f_phase_n30_desync.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $ROOT/gpurun_out/r02/f_smoke.txt 2>&1
python3 bench.py > $ROOT/gpurun_out/r02/f_default_bench.json 2> $ROOT/gpurun_out/r02/f_default_bench.err
tail -c 600 $ROOT/gpurun_out/r02/f_default_bench.json
