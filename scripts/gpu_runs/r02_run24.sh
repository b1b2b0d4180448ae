R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_24
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python scripts/bench_streamk.py 371 > $O/bench_streamk_371.log 2>&1; echo "bench exit $?"; grep "bf16 " $O/bench_streamk_371.log | cut -c1-200
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py tests/test_generate_gpu.py tests/test_train_gpu.py -x -q -p no:cacheprovider > $O/t_llama.log 2>&1; echo "llama tests exit $?"; tail -3 $O/t_llama.log | cut -c1-300
for prec in split bf16; do
timeout 600 python bench.py --stages llama --no-cpu-baseline --llm-precision $prec > $O/bench_llama_$prec.log 2>&1; echo "llama $prec exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_llama_$prec.log | tr '\n' ' ')"
LLARK_STREAMK=0 timeout 600 python bench.py --stages llama --no-cpu-baseline --llm-precision $prec > $O/bench_llama_${prec}_nosk.log 2>&1; echo "llama $prec no-sk exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_llama_${prec}_nosk.log | tr '\n' ' ')"
done
timeout 600 python bench.py --stages generate --no-cpu-baseline > $O/bench_generate.log 2>&1; echo "generate exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' $O/bench_generate.log | tr '\n' ' ')"
LLARK_STREAMK=0 timeout 600 python bench.py --stages generate --no-cpu-baseline > $O/bench_generate_nosk.log 2>&1; echo "generate no-sk exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' $O/bench_generate_nosk.log | tr '\n' ' ')"
