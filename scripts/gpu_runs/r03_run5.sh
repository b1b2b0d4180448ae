#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python scripts/debug/vq_fused_diff.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_vq_fused_diff.txt; cat gpurun_out/r03_vq_fused_diff.txt
