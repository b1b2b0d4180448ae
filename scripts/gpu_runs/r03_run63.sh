#!/bin/bash
# llark_gemm16_t: scheduler's own fragment-read order (shipped) vs reads of sub-step s + 1 pinned ahead of the MFMAs of s (tp1)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_tp1.so timeout 300 python -m pytest tests/test_gemm_tn_gpu.py -q -m gpu -x 2>&1 | tail -2
for m in 4096 2048; do
for v in "" _tp1; do
  echo "=== libllark_hip$v.so M=$m"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip$v.so timeout 200 python scripts/bench_gemm_train.py 200 $m 2>&1 | grep -E "variant 200"
done; done > gpurun_out/r03_gemm_tn_pipe.txt 2>&1
cat gpurun_out/r03_gemm_tn_pipe.txt | cut -c1-110
