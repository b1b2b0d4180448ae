#!/bin/bash
# round 3, GPU run 3: variant 31 with MFMA-first sub-steps and requests / reads spread over the MFMA gaps: bit-identity, A/B vs v30, phase profile
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256" 2>&1 | tail -4 ) > gpurun_out/r03_run3_tests.txt; cat gpurun_out/r03_run3_tests.txt
if grep -q "failed\|error" gpurun_out/r03_run3_tests.txt; then echo "TESTS FAILED"; exit 1; fi
( timeout 600 python scripts/bench_gemm256.py 30,31,41 2>&1 | grep -v "^{\|amdgpu.ids" ) > gpurun_out/r03_gemm256n_ab_v2.txt; cat gpurun_out/r03_gemm256n_ab_v2.txt
