#!/bin/bash
# llark_gemm16_t wide tile: one LDS stage x 2 workgroups per CU (shipped) vs two stages x 1 workgroup of 4 waves (tw1) or of 8 waves (tw2)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for v in "" _tw1 _tw2; do
  echo "=== libllark_hip$v.so"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip$v.so timeout 200 python scripts/bench_gemm_train.py 200 4096 2>&1 | grep -E "dX|dW|sum over"
done > gpurun_out/r03_gemm_tn_stages_m4096.txt 2>&1
cat gpurun_out/r03_gemm_tn_stages_m4096.txt | cut -c1-110
