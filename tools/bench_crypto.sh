#!/bin/bash
# crypto env on one box: the three batch sizes + the rollout segment, optional phase timelines.
#   bash tools/bench_crypto.sh <tag> [timeline]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-c}
O=$ROOT/gpurun_out/${RDIR:-r03}
mkdir -p $O
run() { name=$1; shift; python3 bench.py --no-cpu-baseline --env crypto "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || echo "FAILED $name"; }
run crypto32k --envs-per-gpu 32768 --steps 3000 --warmup 500
run crypto64k --steps 3000 --warmup 500
run crypto256k --envs-per-gpu 262144 --steps 2000 --warmup 500
run crypto32k_rollout --envs-per-gpu 32768 --rollout 16 --steps 3200 --warmup 480
if [ "$2" = "timeline" ]; then
  python3 tools/phase_times_crypto.py 32768 > $O/${TAG}_phase_32k.txt 2>&1
  python3 tools/phase_times_crypto.py 262144 > $O/${TAG}_phase_256k.txt 2>&1
fi
python3 - <<'PY' $O $TAG
import json,sys,glob,os
d,t=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(os.path.join(d,t+'_crypto*.json'))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print('%-22s us/step %8.2f  value %.3e  frac %.3f'%(os.path.basename(f)[len(t)+1:-5], j['roofline']['avg_launch_us'], j['value'], j['roofline']['frac']))
    except Exception as ex:
        print(os.path.basename(f),'ERR',ex, open(f[:-5]+'.err').read()[-400:])
PY
