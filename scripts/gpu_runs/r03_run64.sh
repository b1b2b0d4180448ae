#!/bin/bash
# RMSNorm backward at width 4096: two waves per row, 4-wave workgroups (shipped) vs four waves per row in workgroups of 4 / 8 / 16 waves
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for v in "" _rw4 _rw4n8 _rw4n16; do
  echo "=== libllark_hip$v.so"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip$v.so timeout 200 python -m pytest tests/test_train_gpu.py -q -m gpu -k rmsnorm_bwd 2>&1 | tail -1
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip$v.so timeout 100 python scripts/bench_rmsnorm_bwd.py 2048 4096 8192 2>&1 | grep rmsnorm_bwd
done > gpurun_out/r03_rmsnorm_bwd_wpr.txt 2>&1
cat gpurun_out/r03_rmsnorm_bwd_wpr.txt
