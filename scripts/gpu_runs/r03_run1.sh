#!/bin/bash
# round 3, GPU run 1: the new two-pass 256x256 tile (variant 31) -- correctness first (bounded), then same-process A/B, then the suite + bench
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256 or tile_variants or persistent" 2>&1 | tail -15 ) > gpurun_out/r03_run1_gemm_tests.txt
cat gpurun_out/r03_run1_gemm_tests.txt
if grep -q "failed\|error\|Timeout" gpurun_out/r03_run1_gemm_tests.txt; then echo "GEMM TESTS FAILED -- skipping the rest"; exit 1; fi
( timeout 600 python scripts/bench_gemm256.py 30,31,41 2>&1 | tail -30 ) > gpurun_out/r03_gemm256n_ab.txt
cat gpurun_out/r03_gemm256n_ab.txt
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r03_run1_suite.txt
cat gpurun_out/r03_run1_suite.txt
( timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -3 ) > gpurun_out/r03_bench_e2e_v1.json
cat gpurun_out/r03_bench_e2e_v1.json
