#!/bin/bash
# quick GPU check used during round 2: stock parity tests + N=100 / headline / desync bench lines
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/r02
TAG=${1:-q}
python -m pytest tests/test_gpu_stock_parity.py tests/test_gpu_raw_ctypes_binding.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_configs.py -x -q > gpurun_out/r02/${TAG}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02/${TAG}_tests.log
tail -5 gpurun_out/r02/${TAG}_tests.log
python3 bench.py --no-cpu-baseline --tickers 100 --turbulence-pct 90 --steps 2000 --warmup 500 > gpurun_out/r02/${TAG}_bench_n100.json 2>gpurun_out/r02/${TAG}_bench_n100.err
python3 bench.py --no-cpu-baseline --tickers 100 --turbulence-pct 90 --desync --steps 1000 --warmup 300 > gpurun_out/r02/${TAG}_bench_n100_desync.json 2>>gpurun_out/r02/${TAG}_bench_n100.err
python3 bench.py --no-cpu-baseline --steps 6000 --warmup 2000 > gpurun_out/r02/${TAG}_bench.json 2>gpurun_out/r02/${TAG}_bench.err
python3 bench.py --no-cpu-baseline --desync --steps 2000 --warmup 500 > gpurun_out/r02/${TAG}_bench_desync.json 2>>gpurun_out/r02/${TAG}_bench.err
python3 - <<'PY' $ROOT/gpurun_out/r02 $TAG
import json,sys,glob,os
d,t=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(os.path.join(d,t+'_bench*.json'))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), 'us/launch %.2f'%j['roofline']['avg_launch_us'], 'frac %.3f'%j['roofline']['frac'])
    except Exception as ex:
        print(os.path.basename(f),'ERR',ex)
PY
if [ -f finrl_amd/lib/libfinenv_diag.so ]; then python3 tools/phase_times.py 65536 100 > gpurun_out/r02/${TAG}_phase_n100.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02/${TAG}_phase_n100.txt; python3 tools/phase_times.py 65536 30 > gpurun_out/r02/${TAG}_phase_n30.txt 2>&1; grep -v amdgpu.ids gpurun_out/r02/${TAG}_phase_n30.txt; fi
