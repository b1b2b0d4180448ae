mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/tests18.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests18.log | tail -2; grep -E "^E  |worst grad" gpurun_out/tests18.log | cut -c1-400 | head -20
