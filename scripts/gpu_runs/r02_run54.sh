# round 2, last check: every GPU test, smoke(), default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_54
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/tests_full.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" $O/tests_full.log | tail -2; grep -E "^E  |^FAILED" $O/tests_full.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_e2e.log | tr '\n' ' ')"
