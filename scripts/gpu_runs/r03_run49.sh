#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_tr -name '*.db' | head -1) gpurun_out/r03_train_default_final_kernel_stats.txt; head -34 gpurun_out/r03_train_default_final_kernel_stats.txt | cut -c1-140; rm -rf gpurun_out/r03_trace_tr
