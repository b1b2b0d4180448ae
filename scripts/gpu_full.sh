# full round-end check: every GPU test, smoke(), and the three bench stages
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_full.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests_full.log | tail -2; grep -E "^E  |^FAILED" gpurun_out/tests_full.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' gpurun_out/bench_e2e.log | tr '\n' ' ')"
timeout 600 python bench.py --stages train --no-cpu-baseline > gpurun_out/bench_train.log 2>&1; echo "train exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_train.log | tr '\n' ' ')"
timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/bench_generate_b1.log 2>&1; echo "gen exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/bench_generate_b1.log | tr '\n' ' ')"
timeout 600 python bench.py --stages jukebox --no-cpu-baseline > gpurun_out/bench_jukebox.log 2>&1; echo "jukebox exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_jukebox.log | tr '\n' ' ')"
timeout 600 python bench.py --stages mpt > gpurun_out/bench_mpt.log 2>&1; echo "mpt exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_mpt.log | tr '\n' ' ')"
timeout 600 python bench.py --stages clap --no-cpu-baseline > gpurun_out/bench_clap.log 2>&1; echo "clap exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_clap.log | tr '\n' ' ')"
