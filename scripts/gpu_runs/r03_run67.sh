#!/bin/bash
# decode attention with the first K / V rounds read ahead, RMSNorm backward reading the old dx with the row: parity, then timings
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 60 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py "tests/test_fuzz_gpu.py::test_attention_random_shapes" "tests/test_train_gpu.py::test_rmsnorm_bwd_vs_autograd" -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r03_decode_attn_tests.txt; cat gpurun_out/r03_decode_attn_tests.txt
timeout 20 python scripts/bench_rmsnorm_bwd.py 2048 4096 2>&1 | grep rmsnorm_bwd | tee gpurun_out/r03_rmsnorm_bwd_v2.txt
cd /tmp && timeout 45 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_dec2 -o dec -- python $GRAFT_REPO_ROOT/scripts/bench_decode.py split > $GRAFT_REPO_ROOT/gpurun_out/r03_decode_v2.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_dec2 -name '*.db' | head -1) gpurun_out/r03_decode_v2_kernel_stats.txt; grep -E "attn_decode|gemv_dma|skinny" gpurun_out/r03_decode_v2_kernel_stats.txt | cut -c1-150; grep "decode" gpurun_out/r03_decode_v2.log | head -6; rm -rf gpurun_out/r03_trace_dec2
