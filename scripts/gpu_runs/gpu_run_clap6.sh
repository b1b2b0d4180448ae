#!/bin/bash
mkdir -p gpurun_out/clap
timeout 400 python scripts/bench_clap_gemm.py > gpurun_out/clap/gemm_sweep.txt 2>&1; echo "exit $?"
grep -v amdgpu.ids gpurun_out/clap/gemm_sweep.txt | tail -30
