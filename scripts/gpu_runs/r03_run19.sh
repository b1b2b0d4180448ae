#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python scripts/bench_gemm_train.py 11,12,13,100,101 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm_train_variants_m4096.txt
timeout 600 python scripts/bench_gemm_train.py 11,12,13,100,101 2048 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm_train_variants_m2048.txt
timeout 600 python scripts/bench_gemm_train.py 11,12,13,100,101 2968 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm_train_variants_m2968.txt
