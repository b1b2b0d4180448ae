#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
echo "== NQ=2 (shipped)"; timeout 300 python scripts/bench_attn.py 2>&1 | grep "backward"
echo "== NQ=1"; LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nq1.so timeout 300 python scripts/bench_attn.py 2>&1 | grep "backward"
( LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_nq1.so timeout 600 python -m pytest tests/test_attn_bwd_gpu.py -q 2>&1 | tail -2 )
