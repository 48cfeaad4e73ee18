# A/B of build variants on ONE box (interleaved, 3 rounds): tools/run_variants.sh
for round in 1 2 3; do
for v in u1 u2 u4; do
  echo "== $v (round $round)"; FINENV_LIB=$PWD/finrl_amd/lib/variants/libfinenv_$v.so python tools/sweep_stock.py --envs 65536 --steps 1500 2>&1 | grep -v amdgpu.ids
done; done
