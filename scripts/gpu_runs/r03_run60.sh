#!/bin/bash
# attention backward with the pinned software pipeline: parity tests + micro-benchmark
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_attn_bwd_gpu.py tests/test_mpt_gpu.py -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r03_attn_pipe_tests.txt; cat gpurun_out/r03_attn_pipe_tests.txt
timeout 200 python scripts/bench_attn.py 2>&1 | tail -8 > gpurun_out/r03_bench_attn_v4.txt; cat gpurun_out/r03_bench_attn_v4.txt
