#!/bin/bash
# CLAP HTSAT stage: kernel + engine parity tests only
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_clap_gpu.py -x -q -m gpu > gpurun_out/clap_tests.log 2>&1
echo "exit $?" >> gpurun_out/clap_tests.log
tail -40 gpurun_out/clap_tests.log
