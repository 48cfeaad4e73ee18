#!/bin/bash
# Evidence pass on ONE box: rocprofv3 kernel stats + per-dispatch durations + PMC traffic for every
# step kernel, the un-profiled bench lines, in-kernel phase timelines, smoke, the default bench run.
# Outputs under gpurun_out/<round>/ev_<tag>/ (tools/collect_evidence.py copies summaries to profiles/).
#   bash tools/evidence.sh r03 f
RD=${1:-r03}; TAG=${2:-f}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/$RD/ev_$TAG
mkdir -p $O
P="bash tools/prof_one.sh"
cd $ROOT
$P $O/stock_n30 stock_step
$P $O/stock_n100 stock_step --tickers 100 --turbulence-pct 90
$P $O/stock_desync stock_step --desync
$P $O/portfolio portfolio_step --env portfolio
$P $O/crypto_64k "crypto_kernel<false" --env crypto
$P $O/crypto_32k "crypto_kernel<false" --env crypto --envs-per-gpu 32768
$P $O/crypto_256k "crypto_kernel<false" --env crypto --envs-per-gpu 262144
$P $O/stocknp "stocknp_kernel<false" --env stocknp
$P $O/cashpenalty "cashpenalty_kernel<false" --env cashpenalty
$P $O/stoploss stoploss_step --env stoploss
mkdir -p $O/drivercmd
echo "--- driver command under the kernel trace"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/drivercmd/trace -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drivercmd/trace.log 2>&1 ) ; mkdir -p $O/drivercmd
python3 tools/prof_summary.py $O/drivercmd stock_step > $O/drivercmd/summary.json
echo "--- portfolio: where the bench's microseconds go (VERDICT r02 item 5)"
for v in "default" "--action-pool 1"; do
  python3 bench.py --no-cpu-baseline --env portfolio --steps 1500 --warmup 300 ${v#default} > "$O/portfolio_${v// /_}.json" 2>/dev/null
done
NO_PMC=1 PROF_STEPS=1500 $P $O/portfolio_long portfolio_step --env portfolio
echo "--- un-profiled bench lines"
RDIR=$RD/ev_$TAG bash tools/bench_all.sh bench
RDIR=$RD/ev_$TAG bash tools/bench_crypto.sh bench
python3 bench.py --no-cpu-baseline --env crypto --envs-per-gpu 32768 --rollout 16 --no-graph --steps 3200 --warmup 480 > $O/bench_crypto32k_rollout_eager.json 2>/dev/null
python3 tools/bench_riskpre.py 2>/dev/null | grep shape > $O/riskpre.jsonl
echo "--- timelines"
for a in "phase_times.py 65536 30:n30" "phase_times.py 65536 100:n100" "phase_times.py 65536 30 desync:n30_desync" \
         "phase_times_crypto.py 32768:crypto_32k" "phase_times_crypto.py 262144:crypto_256k" \
         "phase_times_stocknp.py:stocknp" "phase_times_cashpenalty.py:cashpenalty" "phase_times_stoploss.py:stoploss"; do
  python3 tools/${a%%:*} 2>&1 | grep -v amdgpu.ids > $O/phase_${a##*:}.txt
done
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python3 bench.py > $O/default_bench.json 2> $O/default_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drivercmd_bench.json 2> $O/drivercmd_bench.err
tail -c 400 $O/default_bench.json
