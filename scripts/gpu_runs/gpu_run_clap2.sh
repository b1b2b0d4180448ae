#!/bin/bash
# CLAP stage 4: front-end kernel + module tests
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_clap_gpu.py -x -q -m gpu -s > gpurun_out/clap_tests2.log 2>&1
echo "exit $?" >> gpurun_out/clap_tests2.log
grep -v "amdgpu.ids" gpurun_out/clap_tests2.log | tail -40
