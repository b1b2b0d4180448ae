#!/bin/bash
# dW 16x16x32 kernel: residual loads 5 rows ahead (default) vs 1 row (e1); start stagger of the first 512 workgroups (s8, s16); no epilogue (d1).
mkdir -p gpurun_out/r06
out=gpurun_out/r06/dw16_epilogue.txt
: > $out
for tag in "" _b16e1 _b16s8 _b16s16 _b16d1 "" _b16e1; do
  echo "== libllark_hip$tag.so" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip$tag.so timeout 300 python scripts/bench_gemm_train.py 221 4096 2>&1 | grep "variant 221" >> $out
done
timeout 300 python -m pytest tests/test_gemm_tn_gpu.py -x -q -k "16x16x32" 2>&1 | tail -2 >> $out
cat $out
