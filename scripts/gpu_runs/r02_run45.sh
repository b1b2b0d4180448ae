R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_45
mkdir -p $O
export TMPDIR=/tmp
cd $R
export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_skew.so
timeout 600 python -m pytest tests/test_lo8_gpu.py tests/test_prior_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
for v in base skew base skew; do
  if [ $v = base ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  echo "== $v" | tee -a $O/skew.log; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-120 | tee -a $O/skew.log
done
