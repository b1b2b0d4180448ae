#!/bin/bash
# round 4, run 6: gemm256x as the prior's default for K >= 2048 (phase R skipped on the ragged last column tile): GEMM tests, the 36-layer
# parity fixtures with it, A/B vs variant 31, and the e2e step
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_fulldepth_gpu.py -x -q -s -k "gemm256 or jukebox" 2>&1 | grep -E "fulldepth|passed|failed|Error|error" | tail -20 ) > gpurun_out/r04/run6_tests.txt
( timeout 600 python scripts/bench_gemm256.py 31,32 2>&1 | grep -v DIFFERENT | tail -9 ) > gpurun_out/r04/gemm256x_ab_v3.txt
( timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 ) > gpurun_out/r04/run6_bench.txt
cat gpurun_out/r04/run6_tests.txt; head -8 gpurun_out/r04/gemm256x_ab_v3.txt; cut -c1-1800 gpurun_out/r04/run6_bench.txt
