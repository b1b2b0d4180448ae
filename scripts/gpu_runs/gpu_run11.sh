mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/tests11.log 2>&1; echo "tests exit $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"
timeout 600 python bench.py --stages llama --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_llama_split.log 2>&1; echo "bench split exit $?"
timeout 600 python bench.py --stages llama --steps 3 --warmup 1 --no-cpu-baseline --llm-precision bf16 > gpurun_out/bench_llama_bf16.log 2>&1; echo "bench bf16 exit $?"
grep -E "passed|failed" gpurun_out/tests11.log | tail -2; grep -E "^E  |rel err" gpurun_out/tests11.log | cut -c1-330 | head -20; tail -1 gpurun_out/smoke.log; tail -1 gpurun_out/bench_llama_split.log | cut -c1-700; tail -1 gpurun_out/bench_llama_bf16.log | cut -c1-700
