mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fuzz_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/tests56.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests56.log | tail -2; grep -E "^E  |^FAILED" gpurun_out/tests56.log | cut -c1-260 | head -30
