R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_39
mkdir -p $O
export TMPDIR=/tmp
cd $R
for b in 1 2 4; do for fr in 512 129; do
LLARK_FRAG_MIN_ROWS=$fr timeout 600 python bench.py --stages llama --batch $b --no-cpu-baseline > $O/bench_llama_b${b}_fr$fr.log 2>&1; echo "llama b$b frag_min_rows=$fr exit $?: $(grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"llama_ms_per_step": [0-9.]*' $O/bench_llama_b${b}_fr$fr.log | tr '\n' ' ')" | tee -a $O/summary.log
done; done
