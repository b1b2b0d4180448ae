# round 2, GPU run 49: full suite + smoke + bench stages + kernel trace (stream-K Llama GEMMs, tile-overlapped lo8 GEMM, rewritten attention stages, skewed lo8 loop)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_49
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/tests_full.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" $O/tests_full.log | tail -2; grep -E "^E  |^FAILED" $O/tests_full.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' $O/bench_e2e.log | tr '\n' ' ')"
timeout 600 python bench.py --stages jukebox --no-cpu-baseline > $O/bench_jukebox.log 2>&1; echo "jukebox exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_jukebox.log | tr '\n' ' ')"
timeout 600 python bench.py --stages generate --no-cpu-baseline > $O/bench_generate_b1.log 2>&1; echo "gen exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_generate_b1.log | tr '\n' ' ')"
timeout 600 python bench.py --stages train --no-cpu-baseline > $O/bench_train.log 2>&1; echo "train exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_train.log | tr '\n' ' ')"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_e2e.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $O/prof/r02_results.db $O/e2e_kernel_stats.txt; rm -rf $O/prof
head -16 $O/e2e_kernel_stats.txt | cut -c1-170
