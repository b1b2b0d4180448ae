#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemv_dma_gpu.py -x -q 2>&1 | tail -12 ) > gpurun_out/r03_run38_tests.txt; cat gpurun_out/r03_run38_tests.txt
echo "== LDS-DMA GEMV"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== MFMA skinny (LLARK_GEMV_DMA=0)"; LLARK_GEMV_DMA=0 timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
