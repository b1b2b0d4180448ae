mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_infer_driver.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests53.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests53.log | tail -2; grep -E "^E  |Error" gpurun_out/tests53.log | cut -c1-300 | head -20
