#!/bin/bash
# Kernel trace of the train stage with the accumulation fused (8 x 2048 as one pass, 4 loss groups).
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && cd $R
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_trainf -o a -- python bench.py --stages train --batch 8 --micro-batch 8 --loss-groups 4 --grad-comm bf16 --train-seq 2048 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r06/train_fused_trace.log 2>&1
f=$(find gpurun_out/r06/prof_trainf -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r06/train_fused_kernel_stats.txt
rm -rf gpurun_out/r06/prof_trainf
grep "^{" gpurun_out/r06/train_fused_trace.log | tail -1 | cut -c1-400
head -45 gpurun_out/r06/train_fused_kernel_stats.txt | cut -c1-175
