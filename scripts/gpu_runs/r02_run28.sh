R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_28
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in nopre pre nopre pre; do
  if [ $v = pre ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_nopre.so; fi
  echo "== $v"; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-120 | tee -a $O/resid_$v.log
done
unset LLARK_HIP_LIB
timeout 300 python -m pytest tests/test_lo8_gpu.py tests/test_prior_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
