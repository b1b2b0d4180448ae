#!/bin/bash
# round 4, run 7: the whole GPU suite with gemm256x as the prior's default, then the PMC passes of that kernel
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r04/run7_suite.txt
bash scripts/gpu_runs/r04/pmc_gemm256x_full.sh > gpurun_out/r04/run7_pmc.log 2>&1
tail -12 gpurun_out/r04/run7_suite.txt; tail -40 gpurun_out/r04/pmc_gemm256x_full_summary.txt
