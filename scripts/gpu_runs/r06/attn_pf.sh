#!/bin/bash
# Attention forward: K / V^T fragment reads hoisted ahead of their products (ATTN_PREFILL_PF bits: 1 = K, 2 = V^T) and the leaner softmax (ATTN_PREFILL_SM2); sm0 = neither.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/attn_pf.txt
: > $out
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_attn_bwd_gpu.py tests/test_train_gpu.py -x -q -k "attn or prefill or train" 2>&1 | tail -3 >> $out
for tag in _sm0 "" _sm0 ""; do
  echo "== libllark_hip$tag.so" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip$tag.so timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
