R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_52
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base a8at1 w8at3 a1w3 iss1 base; do
  if [ $v = base ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  echo "== $v" | tee -a $O/tune.log; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-100 | tee -a $O/tune.log
done
