#!/bin/bash
# round 4, run 4: gemm256x with the two-half residual epilogue (first half requested ahead of the next tile's prologue): parity + A/B
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256x" 2>&1 | tail -5 ) > gpurun_out/r04/run4_tests.txt
( timeout 600 python scripts/bench_gemm256.py 31,32 2>&1 | grep -v DIFFERENT | tail -9 ) > gpurun_out/r04/gemm256x_ab_v2.txt
tail -3 gpurun_out/r04/run4_tests.txt; head -8 gpurun_out/r04/gemm256x_ab_v2.txt
