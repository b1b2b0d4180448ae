#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_dec -o dec -- python $GRAFT_REPO_ROOT/scripts/bench_decode.py split > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_dec -name '*.db' | head -1) gpurun_out/r03_decode_split_kernel_stats.txt; grep "gemv_dma\|attn_decode\|rmsnorm\|skinny" gpurun_out/r03_decode_split_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_dec
