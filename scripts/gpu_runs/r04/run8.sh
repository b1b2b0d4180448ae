#!/bin/bash
# round 4, run 8: gemm256x with the request addresses hoisted out of the MFMA gaps: parity + A/B against variant 31 (same process)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256x" 2>&1 | tail -3 ) > gpurun_out/r04/run8_tests.txt
( timeout 600 python scripts/bench_gemm256.py 31,32 2>&1 | grep -v DIFFERENT | tail -9 ) > gpurun_out/r04/gemm256x_ab_v4.txt
tail -2 gpurun_out/r04/run8_tests.txt; head -8 gpurun_out/r04/gemm256x_ab_v4.txt
