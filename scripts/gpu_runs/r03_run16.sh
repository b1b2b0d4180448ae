#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_attn_bwd_gpu.py -q 2>&1 | tail -12 ) > gpurun_out/r03_run16_attn.txt; cat gpurun_out/r03_run16_attn.txt
( timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_train_gpu.py tests/test_mpt_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r03_run16_llama.txt; cat gpurun_out/r03_run16_llama.txt
echo "=== NK=1"; timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_bench_attn_nk1.txt
echo "=== NK=2"; LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nk2.so timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_bench_attn_nk2.txt
