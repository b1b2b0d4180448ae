#!/bin/bash
# What bounds the dW loop: the 16x16x32 kernel with parts knocked out (B16_DEBUG bits: 1 epilogue, 2 X^T loads, 4 LDS reads, 8 DMA, 16 barrier).
mkdir -p gpurun_out/r06
out=gpurun_out/r06/dw16_knockout.txt
: > $out
for d in 0 1 2 4 6 7 15 31; do
  lib=$PWD/llark_amd/libllark_hip_b16d$d.so
  [ $d == 0 ] && lib=$PWD/llark_amd/libllark_hip.so
  echo "== B16_DEBUG=$d" >> $out
  LLARK_HIP_LIB=$lib timeout 300 python scripts/bench_gemm_train.py 221 4096 2>&1 | grep "variant 221" >> $out
done
cat $out
