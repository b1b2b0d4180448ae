mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests9.log 2>&1; echo "tests exit $?"
timeout 900 python scripts/bench_gemm.py 0,1,2,3 > gpurun_out/bench_gemm3.log 2>&1; echo "bench_gemm exit $?"
timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?"
grep -E "passed|failed|^E " gpurun_out/tests9.log | cut -c1-300 | tail -5; grep -v "^{" gpurun_out/bench_gemm3.log | grep -v amdgpu.ids | grep -v "^variant"; tail -1 gpurun_out/bench_e2e.log | cut -c1-1200
