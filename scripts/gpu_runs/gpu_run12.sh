mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests12.log 2>&1; echo "tests exit $?"
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e2e -o r01 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_e2e.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py gpurun_out/prof_e2e/r01_results.db gpurun_out/prof_e2e_stats.txt; rm -rf gpurun_out/prof_e2e
grep -E "passed|failed" gpurun_out/tests12.log | tail -2; grep -E "^E  " gpurun_out/tests12.log | cut -c1-300 | head; tail -1 gpurun_out/bench_e2e.log | cut -c1-1500; head -16 gpurun_out/prof_e2e_stats.txt | cut -c1-150
