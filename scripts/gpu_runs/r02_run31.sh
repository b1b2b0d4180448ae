R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_31
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in 0 1 2 3 4; do
  if [ $v = 0 ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_attn_ab$v.so; fi
  echo "== ablate $v (1 no softmax, 2 no PV MFMAs, 3 no QK MFMAs, 4 no output stores)" | tee -a $O/attn_ablate.log
  timeout 200 python scripts/bench_kernels.py attn 2>&1 | grep "^prior_attn" | tee -a $O/attn_ablate.log
done
