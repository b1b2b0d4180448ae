#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_attn_bwd_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r03_run47_tests.txt; cat gpurun_out/r03_run47_tests.txt
timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_bench_attn_tr.txt
