mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py tests/test_llama_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "skinny or golden or infer_with or wrapped" > gpurun_out/tests13.log 2>&1; echo "tests exit $?"
timeout 600 python scripts/bench_kernels.py decode > gpurun_out/bench_decode.log 2>&1; echo "bench decode exit $?"
grep -E "passed|failed" gpurun_out/tests13.log | tail -2; grep -E "^E  " gpurun_out/tests13.log | cut -c1-300 | head; grep decode gpurun_out/bench_decode.log
