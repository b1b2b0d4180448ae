mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > gpurun_out/tests10.log 2>&1; echo "tests exit $?"
timeout 900 python scripts/bench_gemm.py 1,2,3 > gpurun_out/bench_gemm4.log 2>&1; echo "bench_gemm exit $?"
grep -E "passed|failed|^E " gpurun_out/tests10.log | cut -c1-300 | tail -5; grep -v "^{" gpurun_out/bench_gemm4.log | grep -v amdgpu.ids | grep -v "^variant"
