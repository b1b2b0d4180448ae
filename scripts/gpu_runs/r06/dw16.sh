#!/bin/bash
# dW on the 16x16x32 MFMA shape: parity, then the three dW products of a layer against the 32x32x16 form, same box.
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gemm_tn_gpu.py -x -q -k "16x16x32 or dma_loop" 2>&1 | tail -5 > gpurun_out/r06/dw16_tests.txt
cat gpurun_out/r06/dw16_tests.txt
timeout 600 python scripts/bench_gemm_train.py 220,221 4096 2>&1 | grep -v "^M=4096 \(fwd\|dX\)" > gpurun_out/r06/dw16_bench.txt
cat gpurun_out/r06/dw16_bench.txt
