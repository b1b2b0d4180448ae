R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_40
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py tests/test_mpt_gpu.py tests/test_train_gpu.py tests/test_clap_gpu.py tests/test_fuzz_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --stages generate --no-cpu-baseline > $O/bench_generate.log 2>&1; echo "generate exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' $O/bench_generate.log | tr '\n' ' ')"
