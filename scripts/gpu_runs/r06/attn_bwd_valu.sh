#!/bin/bash
# Attention backward: ALiBi as a template parameter + one fma per exponent / dS factor (bwdold = the kernels before).
mkdir -p gpurun_out/r06
out=gpurun_out/r06/attn_bwd_valu.txt
: > $out
timeout 900 python -m pytest tests/test_attn_bwd_gpu.py tests/test_train_gpu.py tests/test_mpt_gpu.py -x -q 2>&1 | tail -3 >> $out
for tag in _bwdold "" _bwdold ""; do
  echo "== libllark_hip$tag.so" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip$tag.so timeout 300 python scripts/bench_attn.py 2>&1 | grep "backward" >> $out
done
cat $out
