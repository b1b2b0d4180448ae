#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemv_dma_gpu.py -x -q 2>&1 | tail -6 ) > gpurun_out/r03_run41_tests.txt; cat gpurun_out/r03_run41_tests.txt
echo "== LDS-DMA GEMV"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_dec -o dec -- python $GRAFT_REPO_ROOT/scripts/bench_decode.py split > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_dec -name '*.db' | head -1) gpurun_out/r03_decode_dma_kernel_stats.txt; grep "gemv_dma\|attn_decode\|rmsnorm" gpurun_out/r03_decode_dma_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_dec
