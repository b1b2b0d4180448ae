#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_lo8_gpu.py -q -k "attention or attn or prior_tiny or five_b or layer" 2>&1 | tail -6 ) > gpurun_out/r03_run31_tests.txt; cat gpurun_out/r03_run31_tests.txt
echo "=== 3 waves/SIMD (spills)"; timeout 300 python scripts/bench_kernels.py attn 2>&1 | grep prior_attn | tee gpurun_out/r03_prior_attn_wpe3.txt
echo "=== 2 waves/SIMD"; LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_wpe2.so timeout 300 python scripts/bench_kernels.py attn 2>&1 | grep prior_attn | tee gpurun_out/r03_prior_attn_wpe2.txt
