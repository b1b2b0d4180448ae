#!/bin/bash
# round 4, run 3: the prior's GEMM tile on v_mfma_f32_16x16x32 (csrc/gemm256x.hip, variant 32): parity, same-process A/B against variant 31
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256" 2>&1 | tail -15 ) > gpurun_out/r04/run3_tests.txt
( timeout 600 python scripts/bench_gemm256.py 31,32 2>&1 | tail -30 ) > gpurun_out/r04/gemm256x_ab.txt
tail -6 gpurun_out/r04/run3_tests.txt; tail -30 gpurun_out/r04/gemm256x_ab.txt
