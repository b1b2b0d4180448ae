#!/bin/bash
# Prior attention, transposed pattern: the chunks of an in-block offset on one XCD (PRIOR_ATTN_P2_XCD, default 1) against the old mapping.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/prior_attn_p2.txt
: > $out
timeout 900 python -m pytest tests/test_prior_gpu.py -x -q -k "factored_attention or prior_layer or full" 2>&1 | tail -2 >> $out
for tag in _p2x0 "" _p2x0 ""; do
  echo "== libllark_hip$tag.so" >> $out
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip$tag.so timeout 300 python scripts/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
