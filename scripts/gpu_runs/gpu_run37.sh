mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_infer_driver.py tests/test_prior_gpu.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests37.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests37.log | tail -2; grep -E "^E  " gpurun_out/tests37.log | cut -c1-300 | head -20
timeout 600 python scripts/bench_kernels.py decode 2>&1 | grep "eager"
LLARK_DECODE_FUSE_NORM_A=0 timeout 600 python scripts/bench_kernels.py decode 2>&1 | grep "eager"
