mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_train_gpu.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests39.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests39.log | tail -2; grep -E "^E  " gpurun_out/tests39.log | cut -c1-300 | head -20
timeout 300 python scripts/bench_gemm.py 20,11 2>&1 | grep -E "resid|down" 
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' gpurun_out/bench_e2e.log
