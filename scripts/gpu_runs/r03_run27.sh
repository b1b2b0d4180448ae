#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python scripts/bench_gemm_train.py 11,12,200 2048 2>&1 | grep -v amdgpu.ids | grep "dX\|dW\|sum" | tee gpurun_out/r03_gemm_tn_m2048.txt
timeout 600 python scripts/bench_gemm_train.py 11,12,200 4096 2>&1 | grep -v amdgpu.ids | grep "dX\|dW\|sum" | tee gpurun_out/r03_gemm_tn_m4096.txt
