#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q 2>&1 | tail -2 ) > gpurun_out/r03_run13_tests.txt; cat gpurun_out/r03_run13_tests.txt
timeout 300 python scripts/bench_gemm256.py 31,41 2>&1 | grep -v "^{\|amdgpu.ids" > gpurun_out/r03_gemm_lo8n_spread.txt; cat gpurun_out/r03_gemm_lo8n_spread.txt | cut -c1-150
