R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_53
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_fulldepth_gpu.py tests/test_llama_gpu.py -x -q -p no:cacheprovider -k "stream_k or llama or fragment" 2>&1 | tail -2
timeout 300 python scripts/bench_streamk.py 2968 2>&1 | grep "split" | grep " o \| down " | grep "tile128x256\|library" | cut -c1-110
timeout 300 python scripts/bench_streamk.py 371 2>&1 | grep "split" | grep " o \| down " | grep "tile128x256\|library" | cut -c1-110
