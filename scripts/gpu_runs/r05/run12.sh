#!/bin/bash
# Round 5, call 12: the DMA loop for PLAIN bf16 operands too (GEMM_BDA=2 build) against the default build (hi + lo only): bit-identity
# test, kernel microbenchmark (variant 0 = registers, 2 = DMA for plain operands), Llama tests on the GEMM_BDA=2 build, Llama stage A/B.
mkdir -p gpurun_out/r05
out=gpurun_out/r05/run12.txt
: > $out
timeout 300 python -m pytest tests/test_prior_gpu.py -q -k "gemm_bda" 2>&1 | tail -3 >> $out
python scripts/bench_gemm_bda.py 2968 5 2>&1 | grep -v amdgpu.ids | grep -v "split.*variant 0" >> $out
LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_bda2.so timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_prior_gpu.py tests/test_mpt_gpu.py -q -k "not ln and not layernorm and not attention" 2>&1 | tail -3 >> $out
for rep in 1 2 3; do
for lib in libllark_hip.so libllark_hip_bda2.so; do
  LLARK_HIP_LIB=$PWD/llark_amd/$lib timeout 300 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_llm']; o=d['roofline_llm_bf16']
print('$lib', 'split ms', d['ms_per_step'], 'gemm frac', r['frac'], '| bf16 ms', o['llama_ms_per_step'], 'gemm frac', o['frac'], 'whole', o['whole_forward_frac'], 'parity', o['parity']['diff_over_max'])" >> $out
done; done
cat $out
