mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_abi.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests31.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests31.log | tail -2; grep -E "^E  " gpurun_out/tests31.log | cut -c1-300 | head -20
timeout 600 python scripts/bench_kernels.py decode 2>&1 | grep decode | tee gpurun_out/bench_decode.log
