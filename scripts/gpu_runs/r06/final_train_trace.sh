#!/bin/bash
# Kernel trace of the train stage at the recipe shape on the final tree (after the attention kernels of section 6 rows 16-18).
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && cd $R
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_train_final -o a -- python bench.py --stages train --batch 8 --micro-batch 2 --train-seq 2048 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06/final_train_trace.log 2>&1
f=$(find gpurun_out/r06/prof_train_final -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r06/final_train_kernel_stats.txt
rm -rf gpurun_out/r06/prof_train_final
grep "^{" gpurun_out/r06/final_train_trace.log | tail -1 | cut -c1-400
head -45 gpurun_out/r06/final_train_kernel_stats.txt | cut -c1-175
