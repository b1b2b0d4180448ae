R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_38
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --stages jukebox --batch 1 --no-cpu-baseline > $O/bench_jukebox_b1.log 2>&1; echo "jukebox b1 exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' $O/bench_jukebox_b1.log | tr '\n' ' ')"
timeout 600 python bench.py --stages llama --batch 1 --no-cpu-baseline > $O/bench_llama_b1.log 2>&1; echo "llama b1 exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"llama_ms_per_step": [0-9.]*' $O/bench_llama_b1.log | tr '\n' ' ')"
