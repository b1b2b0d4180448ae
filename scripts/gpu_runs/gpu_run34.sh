mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_infer_driver.py tests/test_llama_gpu.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests34.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests34.log | tail -2; grep -E "^E  " gpurun_out/tests34.log | cut -c1-300 | head -20
timeout 600 python bench.py --stages generate --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_generate_b1.log 2>&1; echo "gen exit $?"; tail -c 1500 gpurun_out/bench_generate_b1.log
timeout 600 python bench.py --stages generate --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_generate_b8.log 2>&1; echo "gen8 exit $?"; tail -c 700 gpurun_out/bench_generate_b8.log
