R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_43
mkdir -p $O
cd $R
for v in 0 1 2 3 0 3; do
  if [ $v = 0 ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lnnt$v.so; fi
  echo "LN_NT=$v $(timeout 120 python scripts/bench_ln.py 2>&1 | grep layernorm)" | tee -a $O/ln_nt.log
done
