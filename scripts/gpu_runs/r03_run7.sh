#!/bin/bash
# round 3, GPU run 7: full suite + default bench + kernel trace of the default bench (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r03_run7_suite.txt; cat gpurun_out/r03_run7_suite.txt
timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v2.json; cut -c1-600 gpurun_out/r03_bench_e2e_v2.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace -o e2e -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > $GRAFT_REPO_ROOT/gpurun_out/r03_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace -name '*.db' | head -1) gpurun_out/r03_e2e_v1_kernel_stats.txt; head -30 gpurun_out/r03_e2e_v1_kernel_stats.txt | cut -c1-200
rm -rf gpurun_out/r03_trace
