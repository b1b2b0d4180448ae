mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_llama_gpu.py tests/test_extract_gpu.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/tests4.log 2>&1; echo "tests exit $?" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?" >> gpurun_out/summary.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_e2e.log 2>&1; echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.txt
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py gpurun_out/prof_e2e/r01_results.db gpurun_out/prof_e2e_stats.txt; rm -rf gpurun_out/prof_e2e
cat gpurun_out/summary.txt; grep -E "passed|failed" gpurun_out/tests4.log | tail -2; grep -E "^E  |rel err" gpurun_out/tests4.log | cut -c1-300 | head -20; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench_e2e.log; head -14 gpurun_out/prof_e2e_stats.txt | cut -c1-150
