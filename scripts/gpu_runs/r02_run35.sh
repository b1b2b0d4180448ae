R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_35
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -p no:cacheprovider -k "stream_k" 2>&1 | tail -2
timeout 600 python bench.py --stages llama --no-cpu-baseline --llm-precision bf16 > $O/bench_llama_bf16.log 2>&1; echo "llama bf16 exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_llama_bf16.log | tr '\n' ' ')"
LLARK_STREAMK=0 timeout 600 python bench.py --stages llama --no-cpu-baseline --llm-precision bf16 > $O/bench_llama_bf16_nosk.log 2>&1; echo "llama bf16 no-sk exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_llama_bf16_nosk.log | tr '\n' ' ')"
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py tests/test_train_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
