R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_48
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fulldepth_gpu.py tests/test_extract_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > $O/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' $O/bench_e2e.log | tr '\n' ' ')"
