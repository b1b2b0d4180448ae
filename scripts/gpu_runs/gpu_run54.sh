mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpt_gpu.py -q --tb=short -x -p no:cacheprovider -rA > gpurun_out/tests54.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests54.log | tail -2; grep -E "^E  |Error|worst mpt" gpurun_out/tests54.log | cut -c1-400 | head -20
