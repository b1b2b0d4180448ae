# round 2, GPU run 9: full GPU suite with lo8 as the default prior precision, e2e bench (both precisions), kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_9
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/tests_full.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" $O/tests_full.log | tail -2; grep -E "^E  |^FAILED" $O/tests_full.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_e2e_lo8.log 2>&1; echo "e2e lo8 exit $?"; tail -1 $O/bench_e2e_lo8.log | cut -c1-1500
timeout 900 python bench.py --prior-precision f16x2 --no-cpu-baseline > $O/bench_e2e_f16x2.log 2>&1; echo "e2e f16x2 exit $?"; tail -1 $O/bench_e2e_f16x2.log | cut -c1-400
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_e2e.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py $O/prof/r02_results.db $O/e2e_kernel_stats.txt; rm -rf $O/prof
head -16 $O/e2e_kernel_stats.txt | cut -c1-170
