#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_tn_gpu.py -x -q 2>&1 | tail -12 ) > gpurun_out/r03_run26_tests.txt; cat gpurun_out/r03_run26_tests.txt
