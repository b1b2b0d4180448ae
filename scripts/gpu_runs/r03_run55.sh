#!/bin/bash
# kernel trace of the training step at the recipe's sequence length (2 x 2048 x 4 micro-batches) on the shipped library
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_train -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_train -name '*.db' | head -1) gpurun_out/r03_train_2x2048_v7_kernel_stats.txt; head -40 gpurun_out/r03_train_2x2048_v7_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_train
