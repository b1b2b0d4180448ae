mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > gpurun_out/tests7.log 2>&1; echo "tests exit $?"
timeout 900 python scripts/bench_gemm.py 2,3,6,7,8,9 > gpurun_out/bench_gemm2.log 2>&1; echo "bench_gemm exit $?"
grep -E "passed|failed|^E " gpurun_out/tests7.log | cut -c1-300 | tail -5; grep -v "^{" gpurun_out/bench_gemm2.log | grep -v amdgpu.ids
