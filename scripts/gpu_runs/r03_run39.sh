#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 300 python scripts/bench_gemv.py split 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_bench_gemv_split_m1.txt
