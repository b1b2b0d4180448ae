mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or prior_tiny" > gpurun_out/tests5.log 2>&1; echo "tests exit $?"
timeout 600 python scripts/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1; echo "bench_kernels exit $?"
grep -E "passed|failed|^E " gpurun_out/tests5.log | cut -c1-300 | tail -5; grep -v amdgpu.ids gpurun_out/bench_kernels.log
