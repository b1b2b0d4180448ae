R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_51
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_lo8_gpu.py tests/test_prior_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
for v in nopipe pipe nopipe pipe; do
  if [ $v = pipe ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  echo "== $v" | tee -a $O/pipe.log; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | grep resid | sed 's/split f16 //' | cut -c1-120 | tee -a $O/pipe.log
done
