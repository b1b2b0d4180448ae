#!/bin/bash
# round 3, GPU run 4: instruction-placement variants of gemm256n (G256N_SCHED = 0 / 1 / 2), each vs variant 30 in its own process
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for v in 0 1 2; do
  export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_s$v.so
  echo "=== G256N_SCHED=$v"
  timeout 300 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256 and v31" 2>&1 | tail -1
  timeout 300 python scripts/bench_gemm256.py 30,31 2>&1 | grep -v "^{\|amdgpu.ids"
done > gpurun_out/r03_gemm256n_sched.txt 2>&1
cat gpurun_out/r03_gemm256n_sched.txt
unset LLARK_HIP_LIB
( timeout 900 python -m pytest tests/test_vqvae_gpu.py -x -q -s 2>&1 | tail -15 ) > gpurun_out/r03_vqvae_fused_tests.txt; cat gpurun_out/r03_vqvae_fused_tests.txt
