#!/bin/bash
# every bench line of round 2 on one box (no profiler): gpurun_out/r02/<tag>_*.json
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-b}
O=$ROOT/gpurun_out/${RDIR:-r03}
mkdir -p $O
run() { name=$1; shift; python3 bench.py --no-cpu-baseline "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || echo "FAILED $name"; }
run driver --gpus 1 --steps 20 --warmup 5
run stock
run desync --desync --steps 2000 --warmup 500
run n100 --tickers 100 --turbulence-pct 90 --steps 2000 --warmup 500
run portfolio --env portfolio --steps 1500 --warmup 300
run crypto --env crypto --steps 3000 --warmup 500
run crypto32k --env crypto --envs-per-gpu 32768 --steps 3000 --warmup 500
run crypto32k_rollout --env crypto --envs-per-gpu 32768 --rollout 16 --steps 3200 --warmup 480
run crypto32k_rollout_eager --env crypto --envs-per-gpu 32768 --rollout 16 --no-graph --steps 3200 --warmup 480
run stocknp --env stocknp --steps 2000 --warmup 500
run cashpenalty --env cashpenalty --steps 2000 --warmup 500
run stoploss --env stoploss --steps 1500 --warmup 300
python3 - <<'PY' $O $TAG
import json,sys,glob,os
d,t=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(os.path.join(d,t+'_*.json'))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print('%-22s us/step %8.2f  value %.3e  frac %.3f  B %d'%(os.path.basename(f)[len(t)+1:-5], j['roofline']['avg_launch_us'], j['value'], j['roofline']['frac'], j['roofline']['bytes_per_env_step']))
    except Exception as ex:
        print(os.path.basename(f),'ERR',ex, open(f[:-5]+'.err').read()[-400:])
PY
