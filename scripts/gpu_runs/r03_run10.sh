#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
echo "=== default"; timeout 300 python scripts/bench_gemm256.py 30,31 2>&1 | grep "proj_resid\|qkv" | grep -v "^{"
echo "=== no chunk sync for nk < 32"; LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nosync.so timeout 300 python scripts/bench_gemm256.py 30,31 2>&1 | grep "proj_resid\|qkv" | grep -v "^{"
