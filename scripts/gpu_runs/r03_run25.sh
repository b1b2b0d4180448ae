#!/bin/bash
# kernel trace of the default bench (e2e, f16x2 only) and of the default training stage with the final kernels
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_e2e -o e2e -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt-precision --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_e2e -name '*.db' | head -1) gpurun_out/r03_e2e_final_kernel_stats.txt; head -30 gpurun_out/r03_e2e_final_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_e2e
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_tr -name '*.db' | head -1) gpurun_out/r03_train_default_kernel_stats.txt; head -30 gpurun_out/r03_train_default_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_tr
