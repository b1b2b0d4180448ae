mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/tests24.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests24.log | tail -2; grep -E "^E  " gpurun_out/tests24.log | cut -c1-300 | head -20
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?"; tail -c 3000 gpurun_out/bench_e2e.log
timeout 600 python bench.py --stages llama --llm-precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_llama_bf16.log 2>&1; tail -c 1500 gpurun_out/bench_llama_bf16.log
