#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_vqvae_gpu.py -x -q -s 2>&1 | grep "vqvae fused\|passed\|failed\|Error\|error" ) > gpurun_out/r03_vqvae_fused_tests.txt; cat gpurun_out/r03_vqvae_fused_tests.txt
timeout 600 python scripts/bench_vqvae.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_bench_vqvae_v2.txt; cat gpurun_out/r03_bench_vqvae_v2.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace -o vq -- python $GRAFT_REPO_ROOT/scripts/bench_vqvae.py 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace -name '*.db' | head -1) | grep "vq_stage\|codebook_argmin\|resblock_mfma\|conv" | cut -c1-170; rm -rf gpurun_out/r03_trace
for v in 2 3 4; do
  echo "=== G256N_SCHED=$v"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_s$v.so timeout 300 python scripts/bench_gemm256.py 30,31 2>&1 | grep -v "^{\|amdgpu.ids"
done > gpurun_out/r03_gemm256n_sched_v2.txt 2>&1; grep "===\|v31 \|DIFF" gpurun_out/r03_gemm256n_sched_v2.txt | cut -c1-150
