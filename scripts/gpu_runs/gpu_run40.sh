mkdir -p gpurun_out
export TMPDIR=/tmp
for mb in 2 4; do
timeout 600 python bench.py --stages train --micro-batch $mb --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_train_mb$mb.log 2>&1; echo "mb=$mb exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"peak_hbm_gb": [0-9.]*\|"achieved": [0-9.]*' gpurun_out/bench_train_mb$mb.log | tr '\n' ' ')"
done
cd /tmp; R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o r -- python $R/bench.py --stages train --micro-batch 4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_train.log 2>&1; echo "exit $?"
python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/prof_train -name "*.db" | head -1) > $R/gpurun_out/prof_train_stats.txt
head -30 $R/gpurun_out/prof_train_stats.txt | cut -c1-175
rm -rf $R/gpurun_out/prof_train
