#!/bin/bash
# llark_gemm16_t wide tile: one 64-deep LDS stage (shipped) vs a ring of three 32-deep stages in 72 KiB (tw3), both 2 workgroups per CU
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_tw3.so timeout 300 python -m pytest tests/test_gemm_tn_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in "" _tw3; do
  echo "=== libllark_hip$v.so"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip$v.so timeout 200 python scripts/bench_gemm_train.py 200 4096 2>&1 | grep -E "variant 200"
done > gpurun_out/r03_gemm_tn_ring_m4096.txt 2>&1
cat gpurun_out/r03_gemm_tn_ring_m4096.txt | cut -c1-110
