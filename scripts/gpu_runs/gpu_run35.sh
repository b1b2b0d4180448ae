mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_prior_gpu.py -q --tb=short -x -p no:cacheprovider -k "fragment" > gpurun_out/tests35.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests35.log | tail -2; grep -E "^E  " gpurun_out/tests35.log | cut -c1-300 | head -20
timeout 300 python scripts/bench_gemm.py 100,101,102 2>&1 | grep -E "^bf16|proj_resid |qkv_f32" | tee gpurun_out/bench_gemm7.log
timeout 600 python bench.py --stages llama --llm-precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_llama_bf16.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_llama_bf16.log | head -3
timeout 600 python bench.py --stages llama --llm-precision split --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_llama_split.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_llama_split.log | head -3
