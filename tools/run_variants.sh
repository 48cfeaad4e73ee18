for v in n0 n2; do
  echo "== $v"; FINENV_LIB=$PWD/finrl_amd/lib/variants/libfinenv_$v.so python tools/sweep_stock.py --envs 65536 --diag 0,1,3,7 2>&1 | grep -v amdgpu.ids
done
