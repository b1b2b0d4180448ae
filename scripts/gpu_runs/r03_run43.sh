#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemv_dma_gpu.py -q 2>&1 | tail -3 )
echo "== decode"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
