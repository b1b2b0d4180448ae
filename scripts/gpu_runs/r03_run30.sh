#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mpt_gpu.py -q 2>&1 | tail -8 ) > gpurun_out/r03_run30_tests.txt; cat gpurun_out/r03_run30_tests.txt
