mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > gpurun_out/tests17.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|^E " gpurun_out/tests17.log | cut -c1-300 | tail -5
timeout 900 python scripts/bench_gemm.py 2,10,11,12 2>&1 | grep -v "^{" | grep -v amdgpu.ids | grep -v "^variant"
