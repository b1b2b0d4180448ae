R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_23
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -p no:cacheprovider -k "stream_k or fragment_major" > $O/t_sk.log 2>&1; echo "stream-k tests exit $?"; tail -5 $O/t_sk.log | cut -c1-600
timeout 600 python scripts/bench_streamk.py > $O/bench_streamk.log 2>&1; echo "bench exit $?"; grep "M= 2968" $O/bench_streamk.log | grep "split" | grep " o \| down " | cut -c1-200
grep "M=  371" $O/bench_streamk.log | grep "split" | cut -c1-200
timeout 600 python bench.py --stages llama --no-cpu-baseline > $O/bench_llama.log 2>&1; echo "llama exit $?"; tail -1 $O/bench_llama.log | cut -c1-300
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py -x -q -p no:cacheprovider > $O/t_llama.log 2>&1; echo "llama tests exit $?"; tail -3 $O/t_llama.log | cut -c1-300
