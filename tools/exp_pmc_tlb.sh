#!/bin/bash
# address-translation counters of the N = 100 step kernel at two batch sizes (separate passes per counter set)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out/r03/pmc_tlb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for E in 49152 65536; do
  for CTR in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_HIT_sum"; do
    tag=$(echo $CTR | tr ' ' '+')
    FINENV_OBS_PLACEMENT=first rocprofv3 --pmc $CTR --output-format csv -d $O/${E}_$tag -- python3 $ROOT/bench.py --no-cpu-baseline --prewarm 0 --steps 200 --warmup 50 --tickers 100 --turbulence-pct 90 --envs-per-gpu $E > $O/${E}_$tag.log 2>&1
  done
done
python3 - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for d in sorted(glob.glob(os.path.join(O, "*_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f: print(os.path.basename(d), "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "stock_step" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: sorted(v)[len(v) // 2] for k, v in acc.items()})
PY
